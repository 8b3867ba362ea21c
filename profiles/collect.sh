#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   gpurun --timeout 1800 -- 'bash profiles/collect.sh r02'          (second argument "lines": only steps 1 and 4)
# Outputs go to gpurun_out/<tag>_*; `python profiles/summarise.py <tag>` (CPU) then
# copies the summaries into profiles/ and writes profiles/traffic.json.
set -u
TAG=${1:-r06}
ONLY=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
cp emg3d_amd/build_info.json $OUT/${TAG}_build_info.json 2>/dev/null      # which sources the library on this box was built from
{ hostname; /opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI3" | sed 's/^ *//'; date -u +%Y-%m-%dT%H:%MZ; } | tr '\n' ' ' > $OUT/${TAG}_box.txt
run() { # name, rocprof args..., -- bench args
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $OUT/${TAG}_${name}.log 2>&1
  # timelines of the cycle-only runs (tools/r05/gaps.py: busy time and gaps per cycle, by kernel) before the trace is deleted
  if [ "$name" = cycle128 ] || [ "$name" = cycle256 ]; then
    local tr=$(find $OUT/${TAG}_${name} -name '*kernel_trace.csv' | head -1)
    [ -n "$tr" ] && python3 tools/r05/gaps.py "$tr" $([ "$name" = cycle128 ] && echo 6 || echo 3) > $OUT/${TAG}_${name}_timeline.txt 2>&1
  fi
  # isolated-sweep runs: the profiler's average over the TIMED launches only -- the last 360 dispatches of the dominant line-sweep kernel
  # (bench.py --mode sweep: 10 samples x 3 directions x (1 warm-up + 2 timed sweeps) x 4 colour launches); the set-up in front of them
  # launches the same kernel on the placement search's candidate blocks, which are not what the roofline is priced on
  case "$name" in sweep*)
    # (the profiled process's own bench line: its HIP-event launch time goes beside the profiler's average of the SAME launches)
    grep '^{"' $OUT/${TAG}_${name}.log | tail -1 > $OUT/${TAG}_${name}_line.json
    local tr=$(find $OUT/${TAG}_${name} -name '*kernel_trace.csv' | head -1)
    [ -n "$tr" ] && python3 - "$tr" > $OUT/${TAG}_${name}_timed.json <<'PY'
import collections, csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_line_sweep_" in r["Kernel_Name"]]
tot = collections.defaultdict(float)
for r in rows:
    tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
dom = max(tot, key=tot.get)
d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if r["Kernel_Name"] == dom)
last = [x for _, x in d][-360:]
print(json.dumps({"kernel": dom.replace("void ", "").split("(")[0].replace(", ", ","), "launches_of_the_kernel_in_the_run": len(d),
                  "timed_launches": len(last), "average_ms": sum(last) / len(last) * 1e-6, "min_ms": min(last) * 1e-6, "max_ms": max(last) * 1e-6,
                  "source": "rocprofv3 --kernel-trace: the last 360 dispatches of the dominant line-sweep kernel (the sweeps bench.py times)"}))
PY
  ;; esac
  # keep the summaries only (gpurun merges at most 64 MiB back): per-kernel stats and counter values
  find $OUT/${TAG}_${name} -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
  tail -c 2000 $OUT/${TAG}_${name}.log > $OUT/${TAG}_${name}.log.tail; mv $OUT/${TAG}_${name}.log.tail $OUT/${TAG}_${name}.log
}
if [ "$ONLY" = sweeps ]; then   # only the isolated level-0 sweep profiles (steps 2, 2b)
run sweep128 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep128 -- python3 bench.py --mode sweep --no-cpu
run sweep256 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep256 -- python3 bench.py --mode sweep --workload 256V --no-cpu
run sweep128d --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep128d -- python3 bench.py --mode sweep --source dense --no-cpu
run sweep256d --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep256d -- python3 bench.py --mode sweep --source dense --workload 256V --no-cpu
exit 0
fi
# 1. the bench command itself (default workload + the config_256V object + time-to-tolerance solves; without the CPU
#    baseline leg, which launches no kernels): per-kernel time; the dominant kernels' averages must agree with the
#    roofline objects of the bench line
run bench --kernel-trace --stats --output-format csv -d $OUT/${TAG}_bench -- python3 bench.py --no-cpu --no-dense
# 1b. only the timed cycles of the default workload (where a 128^3 F-cycle spends its time)
run cycle128 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cycle128 -- python3 bench.py --steps 6 --warmup 3 --no-cpu --multi 0 --no-256 --no-tol --batch 0 --no-roofline
# 1b'. only cycles of the 256^3 V-cycle (BASELINE configs[2])
run cycle256 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cycle256 -- python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-roofline
# 1c. only batched cycles (8 sources through the same launches)
run batch8 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_batch8 -- python3 tools/batch_cycle.py 128F 8 6
if [ "$ONLY" = all ]; then
# 2. isolated level-0 sweeps (the launches the roofline object is computed from)
run sweep128 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep128 -- python3 bench.py --mode sweep --no-cpu
run sweep256 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep256 -- python3 bench.py --mode sweep --workload 256V --no-cpu
# 2b. the same sweeps with a dense right-hand side (the launch `roofline.frac` is priced on)
run sweep128d --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep128d -- python3 bench.py --mode sweep --source dense --no-cpu
run sweep256d --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sweep256d -- python3 bench.py --mode sweep --source dense --workload 256V --no-cpu
# 3. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
run fetch128 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch128 -- python3 bench.py --mode sweep --no-cpu
run write128 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write128 -- python3 bench.py --mode sweep --no-cpu
run fetch256 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch256 -- python3 bench.py --mode sweep --workload 256V --no-cpu
run write256 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write256 -- python3 bench.py --mode sweep --workload 256V --no-cpu
run fetch128d --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch128d -- python3 bench.py --mode sweep --source dense --no-cpu
run write128d --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write128d -- python3 bench.py --mode sweep --source dense --no-cpu
run fetch256d --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch256d -- python3 bench.py --mode sweep --source dense --workload 256V --no-cpu
run write256d --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write256d -- python3 bench.py --mode sweep --source dense --workload 256V --no-cpu
# 3b. HBM traffic of EVERY kernel of the 256^3 V-cycle (the x<->y transposes among them: HISTORY R6.4)
run fetch256c --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch256c -- python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-roofline
run write256c --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write256c -- python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol --batch 0 --no-roofline
run sq128 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${TAG}_sq128 -- python3 bench.py --mode sweep --no-cpu
fi
# 4. un-profiled bench lines (roofline.traffic is read from profiles/traffic.json of the PREVIOUS summarise.py run)
python3 bench.py > $OUT/${TAG}_bench_128F.json 2> $OUT/${TAG}_bench_128F.err
python3 bench.py --no-cpu --no-256 --no-tol --batch 2,4,8,16 --batch-tune > $OUT/${TAG}_bench_128F_batch.json 2> $OUT/${TAG}_bench_128F_batch.err
python3 bench.py --no-cpu --no-256 --no-tol --batch 0 --multi 3 > $OUT/${TAG}_bench_128F_multi3.json 2> $OUT/${TAG}_bench_128F_multi3.err
python3 bench.py --workload 256V --steps 3 --warmup 3 --no-cpu --no-tol > $OUT/${TAG}_bench_256V.json 2> $OUT/${TAG}_bench_256V.err
# capacity check: 384^3 (a 108 GB handle; not a BASELINE config)
python3 bench.py --workload 384V --steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0 > $OUT/${TAG}_bench_384V.json 2> $OUT/${TAG}_bench_384V.err
# sizes between the powers of two (launch shapes by rounds of waves, HISTORY R5.19) and fields beyond 4 GiB (k_line_sweep_qc<..., BIG>):
# 448^3; 512^3, a 257 GB handle on the 288 GB of the device
timeout 900 python3 bench.py --workload 448V --steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0 > $OUT/${TAG}_bench_448V.json 2> $OUT/${TAG}_bench_448V.err
timeout 900 python3 bench.py --workload 512V --steps 3 --warmup 2 --no-cpu --no-tol --no-256 --multi 0 --batch 0 > $OUT/${TAG}_bench_512V.json 2> $OUT/${TAG}_bench_512V.err
python3 bench.py --ordering lex --steps 1 --warmup 1 --no-cpu --no-256 --no-tol --multi 0 > $OUT/${TAG}_bench_128F_lex.json 2> $OUT/${TAG}_bench_128F_lex.err
tail -c 400 $OUT/${TAG}_bench_128F.json
