# which HIP runtime calls sit between the last kernel of a step and the first of the next?  (GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/gap && mkdir -p /root/repo/gpurun_out/gap
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /root/repo/gpurun_out/gap -o g -- python3 /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
k = list(csv.DictReader(open(glob.glob("gpurun_out/gap/**/*kernel_trace.csv", recursive=True)[0])))
a = list(csv.DictReader(open(glob.glob("gpurun_out/gap/**/*hip_api_trace.csv", recursive=True)[0])))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last steady step boundary: an elementwise add kernel followed (> 20 us later) by k_idwt_fwd_pipe
idx = [i for i in range(1, len(k)) if "k_idwt_fwd_pipe" in k[i]["Kernel_Name"] and "k_idwt_fwd_pipe" not in k[i - 1]["Kernel_Name"]]
i = idx[-3]
t0, t1 = int(k[i - 1]["End_Timestamp"]), int(k[i]["Start_Timestamp"])
print("gap", (t1 - t0) / 1e3, "us between", k[i - 1]["Kernel_Name"][:60], "and", k[i]["Kernel_Name"][:40])
# correlate by dispatch: which API call launched kernel i, and what ran on the host in the 300 us before it
cid = k[i].get("Correlation_Id")
launch = [r for r in a if r.get("Correlation_Id") == cid]
print("launch call:", [(r["Function"], int(r["Start_Timestamp"]) - t0) for r in launch])
ts = int(launch[0]["Start_Timestamp"]) if launch else t1
for r in a:
    s = int(r["Start_Timestamp"])
    if ts - 400000 <= s <= ts + 20000:
        print(f'{(s - t0) / 1e3:9.1f} us  {(int(r["End_Timestamp"]) - s) / 1e3:7.1f} us  {r["Function"]}')
PY
