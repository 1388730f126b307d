#!/usr/bin/env python3
"""profiles/pmc_dominant.json from a tools/profile_bench.sh summary (gpurun_out/prof_<tag>/summary.json): HBM bytes per
launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KiB; gfx950:
FETCH_SIZE x2 for wide coalesced reads), plus the MFMA-busy and LDS figures of the same passes.
  python tools/make_pmc_dominant.py gpurun_out/prof_r04/summary.json conv3x3_wino_r64_kernel r04
Also the proof that the timed kernel executes all of the work: SQ_INSTS_MFMA per launch must equal the algorithmic FLOPs of the
launch x piece products / 2.25 (Winograd; 1.5 for conv_wino_z128.hip) / 32768 (FLOPs of one 32x32x16 MFMA) -- asserted, and stored for bench.py to print."""
import hashlib
import json
import os
import sys

summary, kernel, tag = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.load(open(summary))
rows = {k: v for k, v in d["pmc"].items() if kernel in k}
assert rows, f"no PMC rows for {kernel}"
tot = {}
for v in rows.values():
    for c, x in v.items():
        tot[c] = tot.get(c, 0.0) + x
n = tot["dispatches"]
stats = [v for k, v in d["stats"].items() if kernel in k]
fetch, write = tot["FETCH_SIZE"] * 1024 / n, tot["WRITE_SIZE"] * 1024 / n
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KSRC = {"conv3x3_wino_r64_kernel": ["video-diffusion_amd/csrc/conv_wino_r64.hip", "video-diffusion_amd/csrc/vd_common.h"],
        "conv3x3_wino_z128_kernel": ["video-diffusion_amd/csrc/conv_wino_z128.hip", "video-diffusion_amd/csrc/vd_common.h"]}.get(kernel, [])
# the un-profiled bench line of the same box and command (tools/profile_bench.sh): algorithmic FLOPs per launch, arithmetic mode
bench = json.load(open(os.path.join(os.path.dirname(summary), "bench.json")))
assert bench["roofline"]["kernel"] == kernel, (bench["roofline"]["kernel"], kernel)
pieces = bench["roofline"].get("piece_products", 6)
gain = bench["roofline"].get("direct_over_executed_multiplications", 2.25)       # 2.25: F(2x2,3x3); 1.5: with the folded column transform
mfma_expected = bench["roofline"]["alg_gflop_per_launch"] * 1e9 * pieces / gain / 32768
mfma_measured = tot["SQ_INSTS_MFMA"] / n
assert abs(mfma_measured / mfma_expected - 1) < 1e-4, f"SQ_INSTS_MFMA per launch {mfma_measured} != expected {mfma_expected}: work skipped or duplicated"
out = {
    "kernel": kernel,
    # bench.py withholds `traffic` when these no longer match the tree (a kernel edited after the PMC passes)
    "kernel_sources": {f: hashlib.sha1(open(os.path.join(ROOT, f), "rb").read()).hexdigest() for f in KSRC},
    "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes, tools/profile_bench.sh {tag}) on "
              f"`bench.py --steps 5 --warmup 1`, {int(n)} dispatches of {kernel}",
    "fetch_size_bytes_per_launch": fetch,
    "write_size_bytes_per_launch": write,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE "
                  "exact; both counters are in KiB",
    "hbm_bytes_per_launch": 2 * fetch + write,
    # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
    "mfma_busy_frac": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / tot["GRBM_GUI_ACTIVE"] / 128 if "GRBM_GUI_ACTIVE" in tot else None,
    "avg_launch_us_rocprof": sum(s["total_ns"] for s in stats) / max(sum(s["calls"] for s in stats), 1) / 1e3,
    "lds_bank_conflict_frac": tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"] if tot.get("SQ_LDS_IDX_ACTIVE") else None,
    "valu_insts_per_mfma": tot["SQ_INSTS_VALU"] / tot["SQ_INSTS_MFMA"] if tot.get("SQ_INSTS_MFMA") else None,
    "mfma_insts_per_launch": mfma_measured,
    "mfma_insts_expected_per_launch": mfma_expected,
    "mfma_insts_ratio": mfma_measured / mfma_expected,
    "lib_source_sha": bench.get("lib_source_sha"),
}
json.dump(out, open("profiles/pmc_dominant.json", "w"), indent=1)
print(json.dumps(out, indent=1))
