"""Timing proxy (round 3): the forward program of the U-Net++ step issued node by node on TWO streams -- the deep nodes (maps <= 64 x 64:
latency-bound launches of <= 256 blocks) beside the large-map nodes (bandwidth-bound) -- against the same program on one stream.
Results of the overlapped arm are NOT valid (the epilogue-statistics scratch is shared by concurrent conv -> norm pairs); only its time is.
usage: python tools/experiments/fwd_overlap_probe.py [dtype=bf16] [B=32] [S=256]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multi_task_breast_cancer_amd import nets
from multi_task_breast_cancer_amd.experiment_init import init_multitask_model, init_optimizer
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.synthetic import synthetic_batch
from multi_task_breast_cancer_amd.trainer import FusedTrainStep

DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
NODES = []          # (name, stream, deps, first op, end op)


def graph(plan, x):
    f = nets.UNETPP_FEATURES

    def convolution(inputs, cout, name):
        return plan.conv_cell(inputs, cout, f"{name}.conv.weight", f"{name}.conv.bias", f"{name}.adn.N.weight", f"{name}.adn.N.bias", 0.1, name)

    def two_conv(inputs, cout, name):
        return convolution([convolution(inputs, cout, f"{name}.conv_0")], cout, f"{name}.conv_1")

    def down(t, cout, name, tag=""):
        return two_conv([plan.maxpool(t, f"{name}.pool{tag}")], cout, f"{name}.convs")

    def upcat(t, skips, cout, name, halves=True):
        up_c = t.C // 2 if halves else t.C
        up = plan.convT(t, up_c, 2, f"{name}.upsample.deconv.weight", f"{name}.upsample.deconv.bias", f"{name}.up")
        return two_conv(list(skips) + [up], cout, f"{name}.convs")

    def node(name, stream, deps, fn):
        a = len(plan.fwd_ops)
        r = fn()
        NODES.append((name, stream, deps, a, len(plan.fwd_ops)))
        return r

    regions = plan.pv("final_conv_0_1.weight").shape[0]

    def head(j, t):
        return plan.conv1x1(t, regions, f"final_conv_0_{j}.weight", f"final_conv_0_{j}.bias", f"final_conv_0_{j}")

    M, D = 0, 1         # main: the 256 / 128 maps; side: everything <= 64 x 64 (emission order = the shipped graph's)
    x00 = node("x00", M, [], lambda: two_conv([x], f[0], "conv_0_0"))
    x10 = node("x10", M, [], lambda: down(x00, f[1], "conv_1_0"))
    x01 = node("x01", M, [], lambda: upcat(x10, [x00], f[0], "upcat_0_1", halves=False))
    x20 = node("x20", D, ["x10"], lambda: down(x10, f[2], "conv_2_0"))
    x11 = node("x11", M, ["x20"], lambda: upcat(x20, [x10], f[1], "upcat_1_1"))
    x02 = node("x02", M, [], lambda: upcat(x11, [x00, x01], f[0], "upcat_0_2", halves=False))
    x30 = node("x30", D, [], lambda: down(x20, f[3], "conv_3_0"))
    x21 = node("x21", D, [], lambda: upcat(x30, [x20], f[2], "upcat_2_1"))
    x12 = node("x12", M, ["x21"], lambda: upcat(x21, [x10, x11], f[1], "upcat_1_2"))
    x03 = node("x03", M, [], lambda: upcat(x12, [x00, x01, x02], f[0], "upcat_0_3", halves=False))
    x40 = node("x40", D, [], lambda: down(x30, f[4], "conv_4_0"))
    x31 = node("x31", D, [], lambda: upcat(x40, [x30], f[3], "upcat_3_1"))
    x22 = node("x22", D, [], lambda: upcat(x31, [x20, x21], f[2], "upcat_2_2"))
    x13 = node("x13", M, ["x22"], lambda: upcat(x22, [x10, x11, x12], f[1], "upcat_1_3"))
    x04 = node("x04", M, [], lambda: upcat(x13, [x00, x01, x02, x03], f[5], "upcat_0_4", halves=False))
    o1, o2, o3, o4 = node("heads", M, [], lambda: [head(j, t) for j, t in ((1, x01), (2, x02), (3, x03), (4, x04))])
    pa = node("pa", D, [], lambda: down(x30, f[4], "process_level_3", tag="a"))
    pb = node("pb", D, [], lambda: down(x31, f[4], "process_level_3", tag="b"))

    def cls():
        feat = two_conv([pa, x40, pb], 512, "classifier.0")
        g = plan.gap(feat, "gap")
        h = plan.linear(g, 256, "classifier.3.weight", "classifier.3.bias", True, "fc1")
        n_cls = plan.pv("classifier.5.weight").shape[0]
        return plan.linear(h, n_cls, "classifier.5.weight", "classifier.5.bias", False, "logits")
    logits = node("cls", D, [], cls)
    return logits, [o1, o2, o3, o4]


nets._graph_unetpp = graph
dev = torch.device("cuda:0")
seed_everything(1993)
model = init_multitask_model("MTUNetPlusPlus", 1, 1, 3, deep_supervision=True).to(dev)
model.set_compute(DT)
step = FusedTrainStep(model, init_optimizer(model, "Adam", 1e-4), alpha=0.5)
batch = synthetic_batch(B, S, S, 0, dev)
for _ in range(3):
    step(*batch)
torch.cuda.synchronize()
prog = step._st.programs["fwd"]
nodes = list(NODES)
assert nodes[-1][4] == prog.n and all(nodes[i][4] == nodes[i + 1][3] for i in range(len(nodes) - 1)), "node ranges must tile the program"
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def fwd_serial():
    prog.run()


def fwd_nodes_one_stream():
    for name, s, deps, a, b in nodes:
        prog.run(a, b - a)


EV = {n[0]: torch.cuda.Event() for n in nodes}
need = {d for n in nodes for d in n[2]}


def fwd_two_streams():
    side.wait_stream(main)
    for name, s, deps, a, b in nodes:
        st = side if s else main
        for d in deps:
            st.wait_event(EV[d])
        prog.run(a, b - a, stream=st)
        if name in need:
            EV[name].record(st)
    main.wait_stream(side)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


def per_node():
    out = []
    for name, s, deps, a, b in nodes:
        out.append((name, s, timeit(lambda: prog.run(a, b - a), 10)))
    return out


for r in range(3):
    print(f"round {r}: forward program, one launch list {timeit(fwd_serial):.3f} ms | node by node, one stream {timeit(fwd_nodes_one_stream):.3f} ms | "
          f"two streams {timeit(fwd_two_streams):.3f} ms", flush=True)
pn = per_node()
print("per node (ms, alone): " + "  ".join(f"{n}[{'D' if s else 'M'}] {t:.3f}" for n, s, t in pn))
print(f"sum main {sum(t for n, s, t in pn if not s):.3f} ms, sum side {sum(t for n, s, t in pn if s):.3f} ms")
