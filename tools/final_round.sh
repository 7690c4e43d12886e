# usage (GPU box): bash tools/final_round.sh rNN — everything under profiles/ for a round, in one call on one box
cd $GRAFT_REPO_ROOT
R=${1:-r02}
O=gpurun_out/profiles_$R
bash tools/profile_round.sh $R > /dev/null 2>&1
bash tools/pmc_util.sh $R > /dev/null 2>&1
python bench.py --workload c5 --scaling strong --total-voices 8192 --steps 32 --warmup 8 --no-cpu-baseline --strong-c5-voices 0 > $O/${R}_c5_8192v_bench.json 2>> $O/bench.err
for w in c2 c3 c4 c5; do python bench.py --workload $w --steps 64 --warmup 16 --no-cpu-baseline --strong-c5-voices 0 > $O/${R}_${w}_bench.json 2>> $O/bench.err; done
for v in 2048 4096; do python bench.py --voices $v --steps 64 --warmup 32 --repeats 3 --no-cpu-baseline --strong-c5-voices 0 > $O/${R}_headline_${v}v_bench.json 2>> $O/bench.err; done
python bench.py --superblock 1 --steps 100 --warmup 20 --no-cpu-baseline --strong-c5-voices 0 > $O/${R}_headline_single_block_launches_bench.json 2>> $O/bench.err
# kernel traces of the two bus-chain workloads (the bus kernel is their dominant one) and the ring-stream micro-benchmark (what the memory system
# gives the mid stage's access stream without its arithmetic)
for w in c2 c4; do bash tools/ktrace.sh --workload $w --repeats 3 --steps 64 --warmup 32 > $O/${R}_${w}_kernel_trace.txt 2>&1; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/ringstream/ringstream.bin tools/ringstream/ringstream.hip || { echo "ringstream build failed" >&2; exit 1; }
( cd tools/ringstream && for v in 0 1 2 3 4 5; do ./ringstream.bin 1024 16 $v 4; done; ./ringstream.bin 1024 16 0 2; ./ringstream.bin 1024 16 0 5; ./ringstream.bin 4096 8 0 4 ) > $O/${R}_ringstream.jsonl
bash tools/c3_round.sh c3tmp > /dev/null 2>&1
cp gpurun_out/c3tmp/c3_rocprofv3_kernel_stats.csv $O/${R}_c3_rocprofv3_kernel_stats.csv
cp gpurun_out/c3tmp/c3_pmc_util.csv $O/${R}_c3_pmc_util.csv
bash tools/bus_group_sweep.sh > $O/${R}_bus_group_sweep.txt 2>&1   # C2 / C4 over the bus chain's sequence length
bash tools/pmc_c.sh c5 "VALUBusy" "SQ_INSTS_VALU SQ_INSTS_SALU" > $O/${R}_c5_pmc_util.txt 2>&1
ls -la $O
python - $O $R <<'PY'
import json, sys, os, glob
O, R = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(os.path.join(O, "*_bench*.json"))):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f"{os.path.basename(f):55s} ms/step {d['ms_per_step']:.4f} kernel/block {r['kernel_ms_per_block']:.4f} frac {r['frac']:.3f} value {d['value']/1e9:.2f} G  {r['kernel']}")
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
print(open(os.path.join(O, f"{R}_headline_pmc_traffic.json")).read())
print(open(os.path.join(O, f"{R}_headline_rocprofv3_dominant_kernel.json")).read())
PY
