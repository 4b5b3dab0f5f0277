bash tools/experiments/tp_r4_hard.sh 10
bash tools/experiments/tp_r4_hard_f16.sh 1 2 3 4 5
