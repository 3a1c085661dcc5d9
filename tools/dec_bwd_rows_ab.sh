set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for rows in 384 640 768 1280; do
  OUT=$R/gpurun_out/ab_rows_$rows
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/dec_bwd_rows_ab.py $rows > $OUT/stdout.log 2> $OUT/stderr.log || true
  find $OUT -name '*_kernel_trace.csv' -delete
  echo "rows $rows"
  find $OUT -name '*_kernel_stats.csv' -exec python3 $R/tools/summarize_stats.py {} \; | grep -i "dec_bwd\|cell_wgrad\|dec_fwd\|enc_block_bwd" | cut -c1-60,75-130
done
