# round 6, call 2: the hoisted cell lookup of the pull form (Renderer::cols_pass) and the release build (-DPG_RELEASE), most important first:
#   1 smoke + the whole GPU suite with four workers on the default build
#   2 bench line of the default build and of the release build (steady state), same box
#   3 same-box A/B (tools/gpu/ab_bench.py): round-5 library (if present) / default / release on coinrun, bigfish, bossfight
#   4 SQ_INSTS_VALU / SALU per render wave of the default build (compare profiles/r06_valu_by_phase.txt)
# usage: bash tools/gpu/r6_call2.sh [tag]
TAG=${1:-r6c2}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
if [ -f procgen_amd/csrc/build_rel/libenv.so ]; then
  PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/build_rel python bench.py --no-cpu-baseline --no-host-landed 2>gpurun_out/${TAG}_bench_rel.err | tail -1 > gpurun_out/${TAG}_bench_rel.json; cut -c1-400 gpurun_out/${TAG}_bench_rel.json
  PROCGEN_AMD_LIB_DIR=$R/procgen_amd/csrc/build_rel timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -x 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_rel.log
fi
LIBS=""
for v in build_r05 build build_rel; do [ -f procgen_amd/csrc/$v/libenv.so ] && LIBS=$LIBS,procgen_amd/csrc/$v; done
timeout 900 python tools/gpu/ab_bench.py ${LIBS#,} coinrun,bigfish,bossfight 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f0 -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed > $R/gpurun_out/${TAG}_f0.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f0 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f0.csv 2>&1
grep "render<\|6renderI" $R/gpurun_out/${TAG}_f0.csv | cut -c1-200
