#!/bin/bash
# tree-kernel iteration: parity tests of the spatial kernel + A/B timing against abtest/lib_base.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02tree; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_known_answers.py tests/test_gpu_golden_and_properties.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log | cut -c1-600
LIBS="abtest/lib_base.so $EXTRA_LIBS dart_env_amd/libdart_stepper.so" bash tools/ab_libs.sh "DartHumanWalker-v1 16384 30 32;DartHumanWalker-v1 16384 20 64;DartWalker3d-v1 16384 30 32;DartWalker3d-v1 16384 20 64;DartDog-v1 16384 60 32;DartHalfCheetah-v1 65536 30 32;DartCartPole-v1 65536 100 32;DartReacher3d-v1 65536 100 32"
