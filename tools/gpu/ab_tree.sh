# A/B of the tree-kernel tasks (HumanWalker, Dog, Walker3d) between the in-tree library ("base") and abtest/lib_before.so on one box: bash tools/gpu/ab_tree.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision $3 --envs 16384 --steps 40 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f$3 %.3f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for v in base before; do run $v DartHumanWalker-v1 64; run $v DartHumanWalker-v1 32; run $v DartDog-v1 32; run $v DartWalker3d-v1 32; done; done
