#!/bin/bash
# ONE parametrised GPU session (replaces the per-session tools/gpu_r*_s*.sh scripts of rounds 3-4, which are in the git history):
#   gpurun --timeout N -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
# Every step writes gpurun_out/<tag>_<step>.{log,json,...}; steps run in the order given, each under its own timeout.
#   pytest[=<-k expr>]    python -m pytest tests -m gpu -q [-k expr]            (ARIA_PYTEST_FILES narrows the file list)
#   bench[=<args>]        python bench.py <args>  ->  <tag>_bench.json         (the driver's line when no args)
#   smoke                 __graft_entry__.smoke()
#   prof[=<args>]         rocprofv3 --kernel-trace --stats of the bench command -> kernel_stats_<tag>.txt (tools/gpu_prof_bench.sh)
#   pmc=<t1,t2>           PMC passes of tools/pmc_targets.py targets, separate runs -> <tag>_pmc_<target>_<counter>.csv (tools/gpu_pmc.sh)
#   py=<script.py args>   any probe under tools/ -> <tag>_<script>.json / .err  (commas separate the arguments)
#   lib=<path|restore>    copy an ablation / variant build over aria_amd/libaria_hip.so for the steps that follow (the box's tree is a scratch
#                         copy); `lib=restore` puts the product library back
#   tag=<newtag>          change the output tag for the steps that follow
#   counters=<regex>      rocprofv3 -L filtered -> <tag>_counters.txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=$1; shift
for step in "$@"; do
  name=${step%%=*}; arg=""; [ "$step" != "$name" ] && arg=${step#*=}
  case $name in
    pytest)
      files=${ARIA_PYTEST_FILES:-tests}
      if [ -n "$arg" ]; then ( time timeout 1500 python -m pytest $files -m gpu -q --durations=8 -k "$arg" 2>&1 | tail -30 ) > gpurun_out/${tag}_pytest.log 2>&1
      else ( time timeout 1500 python -m pytest $files -m gpu -q --durations=8 2>&1 | tail -30 ) > gpurun_out/${tag}_pytest.log 2>&1; fi
      grep -E "passed|failed|error" gpurun_out/${tag}_pytest.log | tail -3 ;;
    bench)
      ( time timeout 900 python bench.py ${arg//,/ } > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err ) 2> gpurun_out/${tag}_bench.time
      cut -c1-600 gpurun_out/${tag}_bench.json ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log ;;
    prof)
      bash tools/gpu_prof_bench.sh $tag ${arg//,/ }; head -12 gpurun_out/kernel_stats_$tag.txt | cut -c1-80,112-175 ;;
    pmc)
      bash tools/gpu_pmc.sh $tag ${arg//,/ } > gpurun_out/${tag}_pmc.log 2>&1; ls gpurun_out/${tag}_pmc_*.csv 2>/dev/null | head ;;
    py)
      set -- ${arg//,/ }; script=$1; base=$(basename ${script%.py})
      timeout 600 python tools/${arg//,/ } > gpurun_out/${tag}_$base.json 2> gpurun_out/${tag}_$base.err
      tail -c 1500 gpurun_out/${tag}_$base.json; grep -v amdgpu.ids gpurun_out/${tag}_$base.err | tail -3 ;;
    lib)
      [ -f /tmp/product.so ] || cp aria_amd/libaria_hip.so /tmp/product.so
      if [ "$arg" = restore ]; then cp /tmp/product.so aria_amd/libaria_hip.so; else cp "$arg" aria_amd/libaria_hip.so; fi
      echo "library: $arg ($(stat -c %s aria_amd/libaria_hip.so) bytes)" ;;
    tag) tag=$arg ;;
    counters)
      ( cd /tmp && rocprofv3 -L 2>&1 | grep -i -E "$arg" | cut -c1-300 | sort -u | head -200 ) > gpurun_out/${tag}_counters.txt; wc -l gpurun_out/${tag}_counters.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
