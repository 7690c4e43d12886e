# usage (GPU box): bash tools/ringstream_pmc.sh rNN — calibrates FETCH_SIZE / WRITE_SIZE on the headline mid stage's own access pattern
# (8 bytes per lane, 128-frame sub-chunks, 2 KB granules: MI355X_MICROARCH.md calibrates the gfx950 correction for 16-byte-per-lane streams
# only). tools/ringstream issues that pattern with a KNOWN byte count: variant 0 reads + writes, 4 reads only, 5 writes only. One rocprofv3
# --pmc pass per counter and variant; result -> gpurun_out/profiles_rNN/rNN_ringstream_pmc.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r04}
O=gpurun_out/profiles_$R
mkdir -p $O
B=tools/ringstream/ringstream.bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $B tools/ringstream/ringstream.hip || { echo "ringstream build failed" >&2; exit 1; }
[ -x $B ] || { echo "ringstream.bin missing" >&2; exit 1; }
VOICES=1024; BLOCKS=16
for v in 0 4 5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rs_${v}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rs_${v}_$c -- $B $VOICES $BLOCKS $v 4 > /tmp/rs_${v}_$c.log 2>&1 || { echo "rocprofv3 pass $v $c failed" >&2; tail -5 /tmp/rs_${v}_$c.log >&2; exit 1; }
  done
done
python3 - $VOICES $BLOCKS > $O/${R}_ringstream_pmc.json <<'PY'
import csv, glob, json, sys
voices, blocks = int(sys.argv[1]), int(sys.argv[2])
frames = 1024
def avg(v, c):
    f = glob.glob(f"/tmp/rs_{v}_{c}/**/*counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Kernel_Name"].startswith("void ring_kernel") or "ring_kernel" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)
# known bytes per dispatch: every ring frame (2 x f64) of 12 rings is read once and written once per voice-frame. The reads also touch the
# look-ahead of the vibrato taps (up to 15 frames beyond each 128-frame window, re-read by the next window): between 1.0 and 143/128 of that.
known = 12 * 16 * voices * frames * blocks
out = {"pattern": "tools/ringstream: 1024 workgroups x 12 f64 [frame][2] rings, 8-byte lanes, 128-frame sub-chunks (the headline mid stage's stream)",
       "voices": voices, "blocks_per_dispatch": blocks, "known_read_bytes_per_dispatch": known, "known_read_bytes_upper": known * 143 / 128,
       "known_write_bytes_per_dispatch": known, "variants": {}}
for v, name in ((0, "reads+writes"), (4, "reads only"), (5, "writes only")):
    f, nf = avg(v, "FETCH_SIZE"); w, nw = avg(v, "WRITE_SIZE")
    out["variants"][name] = {"variant": v, "dispatches": nf, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                             "read_factor_lo": (known / (f * 1024)) if v != 5 and f > 0 else None,
                             "read_factor_hi": (known * 143 / 128 / (f * 1024)) if v != 5 and f > 0 else None,
                             "write_factor": (known / (w * 1024)) if v != 4 and w > 0 else None}
rw = out["variants"]["reads+writes"]
out["calibration"] = {"FETCH_SIZE_x": rw["read_factor_lo"], "FETCH_SIZE_x_upper": rw["read_factor_hi"], "WRITE_SIZE_x": rw["write_factor"],
                      "note": "bytes = counter x 1024 x factor for this access pattern; factors from the reads+writes variant (the kernel's mix); "
                              "the guide's 16-byte-per-lane calibration is FETCH x 2, WRITE x 1"}
print(json.dumps(out, indent=1))
PY
cat $O/${R}_ringstream_pmc.json
