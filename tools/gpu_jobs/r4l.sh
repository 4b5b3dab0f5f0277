bash tools/experiments/tp_r4_hard.sh 8 9
