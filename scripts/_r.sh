timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not clustered and not pruning and not staged_search and not every_filter" > gpurun_out/r02_graph_tests.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r02_graph_tests.log
LOGLEVEL=WARNING timeout -k 10 300 python scripts/bench_host_api.py 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
