set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "64 1024 1024" "1 4096 4096" "1 2048 2048" "64 512 512"; do
  tag=$(echo $c | tr ' ' '_')
  SVAMD_FFT_WIDE_1024=1 timeout -k 10 120 rocprofv3 --kernel-trace -d $R/gpurun_out/fftprof/$tag -o t -- python3 $R/profiles/fft_case.py $c > $R/gpurun_out/fftprof_$tag.log 2>&1 || { echo failed $c; tail -5 $R/gpurun_out/fftprof_$tag.log; exit 1; }
  echo "== $c"
  python3 $R/profiles/kernel_avgs.py $(find $R/gpurun_out/fftprof/$tag -name "*.db" | head -1) | head -8
done
