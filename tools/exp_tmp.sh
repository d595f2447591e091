mkdir -p gpurun_out/r02
for N in 1 2 4; do
if [ $N = 1 ]; then
timeout 900 python bench.py --force-distributed --genes 5000 --reads 25000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/bench_s$N.json 2> gpurun_out/r02/bench_s$N.err
else
SHN_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --genes 5000 --reads 25000000 --scaling strong --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/bench_s$N.json 2> gpurun_out/r02/bench_s$N.err
fi
grep -i "error" gpurun_out/r02/bench_s$N.err | tail -2 | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02/bench_s$N.json").read().strip().splitlines()[-1])
    print($N, d["ms_per_step"], d["scaling"], d["config"]["reads_per_gpu"], d["config"]["transcripts_sha256_16"], d["config"]["transcripts"])
    st = d["config"]["host_stage_seconds_per_step_slowest_rank"] or d["config"]["host_stage_seconds_per_step"]
    print({k:round(v,2) for k,v in st.items() if v >= 0.005})
    print("model (sum of slowest-rank stage times):", round(sum(v for k,v in st.items() if not k.startswith("ext.") and not k.startswith("route.")),2))
except Exception as e:
    print("no json", e)
PY
done
