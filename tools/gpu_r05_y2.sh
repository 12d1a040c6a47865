# Round 5, GPU pass Y2: the new default (hipBLASLt behind rocBLAS, set by the Python mirror): config 4 with the default and with the switch off.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in default 0 default 0; do
if [ $v = default ]; then unset ROCBLAS_USE_HIPBLASLT; else export ROCBLAS_USE_HIPBLASLT=$v; fi
timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05y2_c4.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05y2_c4.json").read())
print("hipblaslt=$v c4", round(d["ms_per_step"],2), d.get("evd_stage_ms"))
PY
done
unset ROCBLAS_USE_HIPBLASLT
python - <<'PY'
import time
t0=time.time()
import __graft_entry__ as g
g.smoke()
print("smoke wall", round(time.time()-t0,1), "s")
PY
