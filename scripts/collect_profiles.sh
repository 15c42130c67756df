#!/bin/bash
# Run on the GPU box (via gpurun):  EVA_COMMIT=<short hash> bash scripts/collect_profiles.sh [round tag, default r06]
# bench line (the driver's command) + rocprofv3 kernel trace of the headline (public_ctx.execute(), --no-legs) + PMC traffic
# passes + SQ-counter pass, summaries written under gpurun_out/prof_bench/ in the names they are committed under in
# profiles/.  The counter passes run bench.py --raw-only: the same launch set per group of 32 triples as execute() issues
# (same kernels, same grids — the trace of the headline shows them), without the encrypt / decrypt / verification kernels
# of the headline in the counters' per-class averages.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-legs"
timeout 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
tail -c 400 $OUT/${TAG}_bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py $ARGS --no-cpu-baseline > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --raw-only --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --raw-only --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --raw-only --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1
python $R/scripts/rocprof_summary.py $OUT/trace "$TAG bench.py $ARGS --no-cpu-baseline (the headline: public_ctx.execute() of 32-product programs, N=2^16, L=10, 64 triples/step on 2 issue queues whose launches overlap, so durations are while sharing the GPU; the k_enc_* / k_encrypt / k_dec_* rows are the set-up encryption and the verification decrypt outside the timed region): rocprofv3 --kernel-trace --stats" > $OUT/${TAG}_bench_kernel_trace.md
cp $OUT/trace/*/*kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
python $R/scripts/pmc_sq_summary.py $OUT/pmc_sq "$TAG bench.py --raw-only --steps 2 --warmup 1 (the launch sets of the headline by direct C-ABI calls; a PMC pass serialises the kernels: durations are with the GPU to itself): SQ counters" > $OUT/${TAG}_bench_sq_counters.md
python $R/scripts/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/bench_pmc_traffic.json > /dev/null
python $R/scripts/valu_issue.py $OUT/pmc_sq 32 $OUT/bench_valu_issue.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_batch -- python $R/scripts/prof_legs.py batch 2 > $OUT/trace_batch.log 2>&1
python $R/scripts/rocprof_summary.py $OUT/trace_batch "$TAG dag_batch leg (256 Sobel DAGs, N=2^14, l=5, 2 repetitions): rocprofv3 --kernel-trace --stats" > $OUT/${TAG}_batch_kernel_trace.md
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_harris -- python $R/scripts/prof_legs.py harris 5 > $OUT/trace_harris.log 2>&1
python $R/scripts/rocprof_summary.py $OUT/trace_harris "$TAG dag leg (Harris N=2^15 L=8, resident + host-valuation replays): rocprofv3 --kernel-trace --stats" > $OUT/${TAG}_harris_kernel_trace.md
# ---- r6: the DAG legs the verdict asked evidence for — config 5 (N = 2^16, L = 12), Harris (single and batched), config 4:
# kernel traces, SQ counters (waves in flight, VALU busy) and HBM traffic (FETCH_SIZE / WRITE_SIZE passes) on THIS tree
PMC_SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES"
for leg in c5 harris_batch; do
  reps=5; [ $leg = harris_batch ] && reps=2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$leg -- python $R/scripts/prof_legs.py $leg $reps > $OUT/trace_$leg.log 2>&1
  python $R/scripts/rocprof_summary.py $OUT/trace_$leg "$TAG $leg (scripts/prof_legs.py $leg $reps): rocprofv3 --kernel-trace --stats" > $OUT/${TAG}_${leg}_kernel_trace.md
done
for leg in c5 harris harris_batch batch; do
  reps=2; [ $leg = harris ] && reps=4
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC_SQ --output-format csv -d $OUT/sq_$leg -- python $R/scripts/prof_legs.py $leg $reps > $OUT/sq_$leg.log 2>&1
  python $R/scripts/pmc_sq_summary.py $OUT/sq_$leg "$TAG $leg (scripts/prof_legs.py $leg $reps; a PMC pass serialises the kernels): SQ counters" > $OUT/${TAG}_${leg}_sq_counters.md
done
{
  echo "# $TAG HBM traffic of the DAG legs from the PMC counters (MI355X)"
  echo
  echo "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no other trace domain) of scripts/prof_legs.py,"
  echo "summed over every dispatch of the run by scripts/dag_pmc_totals.py (KiB units; FETCH_SIZE x 2, the gfx950 correction of"
  echo "MI355X_MICROARCH.md) and divided by the executions of the run (c3 = Harris N=2^15 L=8; the set-up encryption, the key uploads' layout"
  echo "kernels and the first-use hoisting tables are in the sums: the per-execution figures are upper bounds)."
  echo
  for spec in "c5 8 8" "c3 20 20" "harris_batch 2 320" "batch 2 1280"; do
    set -- $spec; leg=$1; reps=$2; execs=$3
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pf_$leg -- python $R/scripts/prof_legs.py pmc_$leg $reps > $OUT/pf_$leg.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pw_$leg -- python $R/scripts/prof_legs.py pmc_$leg $reps > $OUT/pw_$leg.log 2>&1
    python $R/scripts/dag_pmc_totals.py $OUT/pf_$leg $OUT/pw_$leg $execs "$leg (prof_legs.py pmc_$leg $reps: $execs executions, resident valuations)"
  done
} > $OUT/${TAG}_dag_pmc_traffic.md 2>&1
rm -rf $OUT/trace_* $OUT/sq_c5 $OUT/sq_harris $OUT/sq_harris_batch $OUT/sq_batch $OUT/pf_* $OUT/pw_* $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
ls $OUT
