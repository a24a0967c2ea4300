#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python tools/check/outliers.py 2000 > $O/outliers.txt 2>&1; cat $O/outliers.txt | tail -8
timeout 900 python -m pytest -q --durations=8 tests/test_baseline_configs_gpu.py::test_config2_gp_50k_rays_depth4 tests/test_devmap_gpu.py::test_randomised_small_scenes tests/test_devmap_gpu.py::test_cloud_filter_sorts_on_the_digits_the_last_insert_needed tests/test_random_variants_gpu.py > $O/run6_tests.log 2>&1; tail -14 $O/run6_tests.log
