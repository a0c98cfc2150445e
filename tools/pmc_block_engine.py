"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --T 1000 --steps 1 --warmup 1 --cpu-steps 0 --no-extras` -> profiles JSON of the
blocked engine: per-kernel HBM-side bytes per launch and per block of B timesteps (gfx950 correction: FETCH_SIZE x 2 for the wide
coalesced reads, MI355X_MICROARCH.md section HBM; WRITE_SIZE exact).

    python tools/pmc_block_engine.py out.json <fetch_dir> <write_dir>
"""
import json, sys
from pmc_summary import summarise

out, fdir, wdir = sys.argv[1:4]
F, W = summarise(fdir, "FETCH_SIZE"), summarise(wdir, "WRITE_SIZE")
per_block = {}
for name in sorted(set(F) | set(W)):
    short = name.replace("void ", "").replace("psmf::", "")
    if not short.startswith("psmf_blk_") or "gram_mfma" in short or short == "psmf_blk_reduce":
        continue            # per-run kernels (first block's Gram) and helpers are not per-block traffic
    f = F.get(name, {}).get("mean_KB", 0.0) * 1024 * 2.0
    w = W.get(name, {}).get("mean_KB", 0.0) * 1024
    per_block[short] = dict(fetch_bytes=f, write_bytes=w, launches=F.get(name, W.get(name))["launches"])
filt = [k for k in per_block if k.startswith("psmf_blk_filter3")][0]
doc = dict(
    note="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 1 --cpu-steps 0 "
         "--T 1000 --no-extras` (d=100000 r=32 f32, blocked engine).  Counter collection serialises kernel dispatches: the library's stream-concurrency "
         "probe sees that and runs one filter launch per block with event hand-off, so every kernel below is launched once per block of 32 timesteps; "
         "bench.py scales to the blocks of a pass.  Bytes per launch; FETCH_SIZE doubled (gfx950: it tallies 128-byte requests at 64 bytes), WRITE_SIZE exact.",
    dominant_kernel="psmf::" + filt,
    traffic_bytes_per_launch=per_block[filt]["fetch_bytes"] + per_block[filt]["write_bytes"],
    all_kernels_bytes_per_block=sum(v["fetch_bytes"] + v["write_bytes"] for v in per_block.values()),
    per_block=per_block, FETCH_SIZE=F, WRITE_SIZE=W)
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("dominant_kernel", "traffic_bytes_per_launch", "all_kernels_bytes_per_block", "per_block")}, indent=1))
