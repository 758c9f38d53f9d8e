python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q -m gpu -k "stage_c or profile or config4" 2>&1 | tail -3
python tools/k3_probe.py 10000000 10000 500
python tools/k3_probe.py 12500000 2000 2000
MG_LIB_PATH=metalign_amd/libmetalign_hip_phases.so python tools/k3_phases.py
