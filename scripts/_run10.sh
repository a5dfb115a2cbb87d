mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/t10.log 2>&1; tail -5 gpurun_out/t10.log
timeout 200 python bench.py --steps 40 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d > gpurun_out/b10.json 2> gpurun_out/b10.err; echo "rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/b10.json'));print(d['value'],d['ms_per_step'],d['latency_ms']['gpu_frame_chain_p50'],d['roofline']);print(d['stages_ms_per_step'])"
