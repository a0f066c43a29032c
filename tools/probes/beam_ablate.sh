# rocprof average of beam_step_kernel for the in-tree library and each ablated build in build_prof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in base $R/build_prof/libjlm_hip_[A-Z]*.so; do
  if [ "$f" = base ]; then unset JLM_HIP_LIB; else export JLM_HIP_LIB=$f; fi
  rm -rf $R/gpurun_out/prof_b; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b -o run -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 -c "import csv,sys; [print(sys.argv[1], r[\"Name\"][:28], r[\"Calls\"], float(r[\"AverageNs\"])/1e3) for r in csv.DictReader(open(sys.argv[2])) if \"beam_step\" in r[\"Name\"]]" $(basename $f) $R/gpurun_out/prof_b/run_kernel_stats.csv
done
