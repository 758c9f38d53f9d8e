#!/bin/bash
# Runs on the GPU box (gpurun): what the dense-table K1 (configs[3] shapes at world 1, the one-k kernel against the resident index) spends on
# its look-ups: the kernel alone with the flush ablated step by step (resident_ablate knob: CandSink::flush, mg_sketch_kernel.h).
export MG_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
OUT=gpurun_out/k1_dense_ablation.txt; : > $OUT
p=29720
for knob in "" "--knob resident_ablate=3" "--knob resident_ablate=2" "--knob resident_ablate=1"; do
  p=$((p+1)); export MASTER_PORT=$p
  timeout 900 python3 bench.py --gpus 1 --config 3 --steps 6 --warmup 2 --no_cpu_baseline --no_secondary --no_definitions $knob > gpurun_out/abl.json 2> gpurun_out/abl.err
  python3 - "$knob" >> $OUT <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abl.json").read().strip().splitlines()[-1])
print("%-24s K1 alone %.2f ms   pass %.2f ms   kernels %s" % (sys.argv[1] or "(as shipped)", d["roofline"]["kernel_ms_alone"], d["ms_per_step"],
      {k: round(v["ms_per_pass"], 2) for k, v in d["kernel_ms_per_pass"].items()}))
PY
done
cat $OUT
