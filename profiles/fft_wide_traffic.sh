# HBM traffic of the three passes (batch 64 of 1024 x 1024): counters in passes of their own.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -s KILL 120 rocprofv3 --pmc $pmc -d $R/gpurun_out/fftpmc/$pmc -o p -- python3 $R/profiles/fft_case.py 64 1024 1024 3 > $R/gpurun_out/fftpmc_$pmc.log 2>&1 || { echo failed $pmc; tail -5 $R/gpurun_out/fftpmc_$pmc.log; exit 1; }
  echo "== $pmc"
  python3 $R/profiles/pmc_summary.py $(find $R/gpurun_out/fftpmc/$pmc -name "*.db" | head -1) wide_
done
