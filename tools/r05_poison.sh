set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05w
A="--workload T --steps 3 --warmup 1 --views-per-gpu 2 --view-streams 2 --ring 16 --no-train-step --no-extras --no-cpu-baseline --views 1 --no-batch-backward"
for pat in -1 0xA5 0x67 0xFF; do
  timeout -k 10 300 python tools/poison_run.py $pat -- $A --dump-grad gpurun_out/r05w/g_$pat.pt > gpurun_out/r05w/out_$pat.json 2> gpurun_out/r05w/err_$pat.txt; echo "pattern $pat rc=$?"; grep -c implausible gpurun_out/r05w/err_$pat.txt; tail -3 gpurun_out/r05w/err_$pat.txt | cut -c1-300
done
python - <<'PY'
import torch
ref=torch.load("gpurun_out/r05w/g_-1.pt")
for pat in ("0xA5","0x67","0xFF"):
    try:
        g=torch.load(f"gpurun_out/r05w/g_{pat}.pt")
        r=ref if torch.is_tensor(ref) else ref["grad"] if isinstance(ref,dict) else ref
        gg=g if torch.is_tensor(g) else g["grad"] if isinstance(g,dict) else g
        print(pat, "max abs diff", float((gg-r).abs().max()), "ref max", float(r.abs().max()))
    except Exception as e:
        print(pat, "ERR", e)
PY
