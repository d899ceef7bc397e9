#!/bin/bash
# The round's rocprofv3 evidence, run on the GPU box (gpurun):  bash scripts/run_profiles.sh r02
# Each counter set is its own pass with --kernel-trace only (never combined with sys/hip/hsa
# traces); the profiled program is python3 itself.  Raw output -> gpurun_out/prof_<tag>*/, then
# condensed into profiles/ by scripts/summarize_rocprof.py and scripts/summarize_variants.py
# (run those in the build container afterwards, or here).
set -e -o pipefail
TAG=${1:-r04}
PART=${2:-all}   # all | f64 | f32: the float32 passes can run in a call of their own
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B=gpurun_out/prof_${TAG}
V=gpurun_out/prof_${TAG}v
mkdir -p $B $V
if [ "$PART" != "f32" ]; then
BENCH="python3 bench.py --steps 5 --warmup 1 --no-extras --cpu-seconds 0"
echo "== bench: kernel trace + stats"; rocprofv3 --output-format csv --kernel-trace --stats -d $B/trace -o run -- $BENCH > $B/bench_trace.log 2>&1
echo "== bench: FETCH_SIZE";           rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $B/pmc_fetch -o run -- $BENCH > $B/bench_pmc_fetch.log 2>&1
echo "== bench: WRITE_SIZE";           rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $B/pmc_write -o run -- $BENCH > $B/bench_pmc_write.log 2>&1
VAR="python3 scripts/profile_variants.py --nt ${NT:-120} --nt-out ${NT_OUT:-24} --reps 2"
echo "== variants: plain";             $VAR --plan-out $V/plan.json > $V/plain.log 2>&1
echo "== variants: kernel trace";      rocprofv3 --output-format csv --kernel-trace --stats -d $V/trace -o run -- $VAR > $V/trace.log 2>&1
echo "== variants: FETCH_SIZE";        rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $V/pmc_fetch -o run -- $VAR > $V/pmc_fetch.log 2>&1
echo "== variants: WRITE_SIZE";        rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $V/pmc_write -o run -- $VAR > $V/pmc_write.log 2>&1
echo "== variants: SQ";                rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $V/pmc_sq -o run -- $VAR > $V/pmc_sq.log 2>&1
# raw CSVs of the counter passes are large (one row per dispatch and counter): keep what the
# summaries need
find $B $V -name "*_agent_info.csv" -delete
python3 scripts/summarize_rocprof.py $B profiles/${TAG}
python3 scripts/summarize_variants.py $V profiles/${TAG}
echo "profiles done (f64)"
fi
if [ "$PART" = "f64" ]; then exit 0; fi
# --- float32 theta/S (BASELINE.json configs[4]): the same variants, half the bytes per cell
W=gpurun_out/prof_${TAG}v32
mkdir -p $W
VAR32="python3 scripts/profile_variants.py --nt ${NT:-120} --nt-out ${NT_OUT32:-48} --reps 2 --dtype f32"
echo "== f32 variants: plain";        $VAR32 --plan-out $W/plan.json > $W/plain.log 2>&1
echo "== f32 variants: kernel trace"; rocprofv3 --output-format csv --kernel-trace --stats -d $W/trace -o run -- $VAR32 > $W/trace.log 2>&1
echo "== f32 variants: FETCH_SIZE";   rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $W/pmc_fetch -o run -- $VAR32 > $W/pmc_fetch.log 2>&1
echo "== f32 variants: WRITE_SIZE";   rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $W/pmc_write -o run -- $VAR32 > $W/pmc_write.log 2>&1
echo "== f32 variants: SQ";           rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $W/pmc_sq -o run -- $VAR32 > $W/pmc_sq.log 2>&1
find $W -name "*_agent_info.csv" -delete
python3 scripts/summarize_variants.py $W profiles/${TAG}_f32
echo "f32 profiles done"
