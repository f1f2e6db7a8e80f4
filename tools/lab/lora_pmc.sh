#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the fused RMSNorm / SwiGLU + adapter kernels -> gpurun_out/<tag>/lora_pmc.txt
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
for mode in rmslora swilora; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$mode$c
    rocprofv3 --pmc $c -d /tmp/pmc_$mode$c -o p --output-format csv -- python3 tools/kernel_bench.py $mode --B 64 --S 2048 --iters 2 > $OUT/pmc_${mode}_$c.log 2>&1
  done
done
python3 - <<'PY' > $OUT/lora_pmc.txt
import csv, glob, collections
M = 131072
alg = {"rms_lora_kernel<3": (M*1024*2 + 3*M*128, M*1024*2 + M*48*2), "rms_lora_kernel<2": (M*1024*2 + 2*M*128, M*1024*2 + M*32*2),
       "swiglu_lora_kernel": (M*6144*2 + M*384, M*3072*2 + M*16*2), "swiglu_fwd_kernel": (M*6144*2, M*3072*2), "rms_fwd_kernel": (M*1024*2, M*1024*2),
       "lora_project_kernel<1": (M*3072*2 + M*384, M*16*2)}
print("kernel | launches | FETCH_SIZE x2 per launch (GB) | algorithmic reads (GB) | WRITE_SIZE per launch (GB) | algorithmic writes (GB)")
for mode in ("rmslora", "swilora"):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = collections.defaultdict(list)
        for f in glob.glob(f"/tmp/pmc_{mode}{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c: d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        vals[c] = d
    for k in vals["FETCH_SIZE"]:
        key = next((a for a in alg if a in k), None)
        if key is None: continue
        f = vals["FETCH_SIZE"][k]; w = vals["WRITE_SIZE"].get(k, [0.0])
        print(f"{key} ({mode}) | {len(f)} | {2*sum(f)/len(f)*1024/1e9:.3f} | {alg[key][0]/1e9:.3f} | {sum(w)/len(w)*1024/1e9:.3f} | {alg[key][1]/1e9:.3f}")
PY
cat $OUT/lora_pmc.txt
