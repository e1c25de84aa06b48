"""profiles/traffic.json from a PMC summary (tools/pmc_run.sh): HBM bytes per launch of the dominant kernels =
2 x FETCH_SIZE (gfx950: the counter reports half the bytes of 16-byte streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE,
both in KB, stamped with the hash of the kernel sources they were measured on (bench.py drops a stale figure).
    python tools/pmc_traffic.py gpurun_out/pmc_summary_<tag>.txt profiles/traffic.json"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

src, dst = sys.argv[1], sys.argv[2]
cur, data, sha = None, {}, None
for line in open(src):
    if line.startswith("src_sha16="):
        sha = line.strip().split("=", 1)[1]
        continue
    m = re.match(r"\s+(\S+)\s+launches\s+(\d+)\s+avg\s+([\d.]+)", line)
    if m and cur:
        data.setdefault(cur, {})[m.group(1)] = float(m.group(3))
    elif not line.startswith(" ") and not line.startswith("=="):
        cur = line.strip()
out = {}
for key, pat in (("main_fwd_kernel", "main_fwd_kernel"), ("main_bwd_sem_kernel", "main_bwd_sem_kernel"), ("main_bwd_rgb_kernel", "main_bwd_rgb_kernel"),
                 ("main_bwd_base_kernel", "main_bwd_base_kernel")):
    for k, d in data.items():
        if pat in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out[key] = dict(hbm_bytes_per_launch=(2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024, fetch_kb=d["FETCH_SIZE"], write_kb=d["WRITE_SIZE"],
                            source=f"profiles/{os.path.basename(src)} (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)")
            break
# the two long kernels of the main table backward (F = 2 table at cfg 2): WRITE_SIZE needs no calibration; FETCH_SIZE is reported as
# the counter gives it (records are read as 16-byte streams like the rows above, the table's p / m / v likewise: the x2 rule is applied)
for key, pat in (("bin_kernel", "bin_kernel<2"), ("accumulate_kernel", "accumulate_kernel<2")):
    for k, d in data.items():
        if pat in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out[key] = dict(hbm_bytes_per_launch=(2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024, fetch_kb=d["FETCH_SIZE"], write_kb=d["WRITE_SIZE"],
                            hbm_write_bytes_per_launch=d["WRITE_SIZE"] * 1024,
                            source=f"profiles/{os.path.basename(src)} (2 x FETCH_SIZE + WRITE_SIZE; the write side needs no calibration)")
            break
if sha is None:  # summaries of earlier rounds carry no stamp: the figure cannot be tied to a source state
    sha = "unstamped"
note = ("hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB.  The x2 on FETCH_SIZE is MI355X_MICROARCH.md's gfx950 correction for "
        "16-byte-per-lane STREAMING reads; it is calibrated for exactly that access shape.  The four kernels listed here (the main field's "
        "forward and its three backward kernels) read their operands -- feature planes, kept activations -- as 16-byte / lane streams, so the "
        "rule applies.  bin_kernel / accumulate_kernel (main table backward) stream records and p / m / v as 16-byte loads and are listed too; "
        "their WRITE_SIZE needs no calibration (hbm_write_bytes_per_launch).  The gather kernels (grid_encode_*) are deliberately NOT listed: "
        "for 4..16-byte rows fetched as 64-byte lines the factor is uncalibrated; their rooflines in bench.py use algorithmic bytes and "
        "measured line rates instead.")
json.dump(dict(src_sha16=sha, note=note, kernels=out), open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
