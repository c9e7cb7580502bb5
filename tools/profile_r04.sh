export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04prof; mkdir -p $O; cd /tmp
B="--in-flight 1 --no-streaming --no-cpu-baseline --no-c4c5"
rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $R/bench.py $B --steps 5 --warmup 2 > $O/bench_line_profiled.json 2> $O/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -- python3 $R/bench.py $B --steps 2 --warmup 1 --no-profile > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -- python3 $R/bench.py $B --steps 2 --warmup 1 --no-profile > $O/pw.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pm -- python3 $R/bench.py $B --steps 2 --warmup 1 --no-profile > $O/pm.log 2>&1
cd $R
DB=$(find $O/kt -name "*results.db" | head -1); python3 tools/summarize_trace.py $DB > $O/bench_kernel_by_shape.txt; python3 tools/kernel_stats_csv.py $DB $O/bench_kernel_stats.csv
python3 tools/pmc_traffic.py $(find $O/pf -name "*results.db" | head -1) $(find $O/pw -name "*results.db" | head -1) $O/pmc_hbm_traffic.json
python3 tools/pmc_by_kernel.py $O/pm > $O/pmc_mfma_busy.txt
rm -rf $O/kt $O/pf $O/pw $O/pm
python3 bench.py --steps 20 --warmup 5 > $O/bench_line_driver_cmd.json 2> $O/bench_driver.err
ls -la $O; head -12 $O/pmc_mfma_busy.txt | cut -c1-200
# in-kernel timelines (dev builds made on the build box: tools/x3_variant.sh p3st "-DPFHIP_P3_STAMPS=1" gemm_p3.hip; attpst "-DPFHIP_ATTP_STAMPS=1" attention_p3.hip)
if [ -f build/libpfhip_p3st.so ]; then
  PFHIP_LIB=$R/build/libpfhip_p3st.so python3 tools/p3_stamps.py qkv 2>&1 | grep -v "amdgpu.ids\|Warning\|print" > $O/p3_stamps_qkv.txt
  PFHIP_LIB=$R/build/libpfhip_p3st.so python3 tools/p3_stamps.py out 2>&1 | grep -v "amdgpu.ids\|Warning\|print" > $O/p3_stamps_out_projection.txt
fi
if [ -f build/libpfhip_attpst.so ]; then
  PFHIP_LIB=$R/build/libpfhip_attpst.so python3 tools/att_stamps.py 2>&1 | grep -v "amdgpu.ids\|Warning\|print" > $O/attention_p3_stamps.txt
fi
python3 tools/att_probe.py 2>&1 | grep "Lq=" > $O/attention_probe.txt
python3 tools/p3_probe.py 2>&1 | grep -v amdgpu.ids > $O/p3_probe.txt
if [ -f build/libpfhip_p3base.so ]; then PFHIP_LIB=$R/build/libpfhip_p3base.so python3 tools/p3_probe.py 2>&1 | grep -v amdgpu.ids >> $O/p3_probe.txt; fi
ls -la $O
