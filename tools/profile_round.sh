#!/bin/bash
# Profiles of the bench workload for profiles/ (run on the GPU box: gpurun -- tools/profile_round.sh r01_e).
# Kernel trace + stats in one run; every PMC group in its own run with --kernel-trace only
# (rocprofv3 must not combine --pmc with sys/hip/hsa tracing on this pool).
set -e
TAG=${1:-r01_x}; shift || true
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg $*"
run() { name=$1; shift; rocprofv3 "$@" -d "$OUT/$name" -- python bench.py $ARGS > "$OUT/$name.log" 2>&1 || true; find "$OUT/$name" -name '*.db' | head -1; }
db=$(run stats --kernel-trace --stats)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (durations in us)"; python tools/rocpd_summary.py stats "$db"; } > "$OUT/${TAG}_kernel_stats.txt"
grep '^{"metric"' "$OUT/stats.log" | tail -1 > "$OUT/${TAG}_bench_line.json"
{
  echo "# rocprofv3 --kernel-trace --pmc <group> -- python bench.py $ARGS ; one run per group; per-dispatch means"
  echo "# gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md, HBM section)"
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
    n=$(echo "$grp" | tr ' ' '_' | cut -c1-24)
    db=$(run "pmc_$n" --kernel-trace --pmc $grp)
    echo; echo "## --pmc $grp"
    [ -n "$db" ] && python tools/rocpd_summary.py pmc "$db" | grep -v "rocclr\|^void " || echo "(no output)"
  done
} > "$OUT/${TAG}_pmc.txt"
python - "$OUT" "$TAG" <<'PY'
import json, re, sys
out, tag = sys.argv[1], sys.argv[2]
txt = open("%s/%s_pmc.txt" % (out, tag)).read()
def table(counter):
    m = re.search(r"## --pmc %s\n(.*?)(\n\n|\Z)" % counter, txt, re.S)
    d = {}
    if m:
        for line in m.group(1).splitlines()[1:]:
            p = line.split()
            if len(p) >= 3: d[p[0]] = float(p[2])
    return d
f, w = table("FETCH_SIZE"), table("WRITE_SIZE")
def counters(group_first):
    """per-kernel dict of every counter of the PMC group whose header line starts with `group_first`"""
    m = re.search(r"## --pmc %s[^\n]*\n(.*?)(\n\n|\Z)" % group_first, txt, re.S)
    d = {}
    if m:
        rows = m.group(1).splitlines()
        names = rows[0].split()[2:]
        for line in rows[1:]:
            p = line.split()
            if len(p) >= 2 + len(names): d[p[0]] = dict(zip(names, [float(x) for x in p[2:2 + len(names)]]))
    return d
inst = counters("SQ_INSTS_VALU")
line = json.loads(open("%s/%s_bench_line.json" % (out, tag)).read())
cfg = line["config"]
res = {"_comment": "HBM traffic per dispatch from the rocprofv3 PMC passes of %s_pmc.txt: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated). bench.py reports it as roofline.traffic only when its workload matches." % tag,
       "workload": {"width": cfg.get("width", 1920), "height": cfg.get("height", 1080), "features": cfg.get("features", 2000), "batch": cfg["batch"]},
       "scene": cfg.get("scene", "textured"),
       "kernels": {k: dict({"fetch_kb": f[k], "write_kb": w.get(k, 0.0)},
                           **{n.lower(): v for n, v in inst.get(k, {}).items() if n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")}) for k in f}}
json.dump(res, open("%s/%s_traffic.json" % (out, tag), "w"), indent=2)
PY
cat "$OUT/${TAG}_kernel_stats.txt"; cat "$OUT/${TAG}_pmc.txt"; cat "$OUT/${TAG}_bench_line.json"
# the rocprofv3 databases are large (gpurun copies back at most 64 MiB): keep the summaries only
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
