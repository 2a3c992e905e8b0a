# rocprofv3 --pmc passes for the row-window angle kernel (k_angle64_rows, 12 x 63 x 100 planes = 6300 bins): HBM-side traffic.
# Run on the GPU box from the repo root: bash tools/pmc_angle.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/ang_fetch -o p -- python3 tools/angle_shape.py 6300 > /dev/null 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/ang_write -o p -- python3 tools/angle_shape.py 6300 > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import glob, json, sqlite3
res = {}
for kind in ("fetch", "write"):
    for path in glob.glob(f"gpurun_out/prof/ang_{kind}/**/*results.db", recursive=True):
        db = sqlite3.connect(path)
        for kernel, counter, n, total in db.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) "
                                                    "from counters_collection group by kernel_name, counter_name"):
            if "k_angle64_rows" in kernel:
                res[counter] = total / n
                res["kernel"] = kernel.split("(")[0].replace("void ", "")
frames, V, bins, A = 998, 12, 6300, 64
live = V - 2
alg = frames * (V + A) * bins * 8
must = frames * (live + A) * bins * 8
f = res["FETCH_SIZE"]; w = res["WRITE_SIZE"]
fetch = 2 * (1024 * f if f < alg / 512 else f)
write = 1024 * w if w < alg / 512 else w
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH doubled per the gfx950 note) around tools/angle_shape.py 6300: "
               "mmw_angle_fft on 998 frames of 12 x 6300-bin range-Doppler planes, per launch; 'must move' = the 10 live planes in, 64 out",
       "kernel": res["kernel"], "frames": frames, "algorithmic_bytes": alg, "bytes_it_must_move": must,
       "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_over_must_move": (fetch + write) / must,
       "fetch_over_live_input": fetch / (frames * live * bins * 8), "write_over_output": write / (frames * A * bins * 8)}
json.dump(out, open("gpurun_out/prof/angle_rows_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
