# round 6, call 3: the display list (pg_prep.h: prep -> raster -> render_list) on the device, most important first:
#   1 smoke, the display-list tests, coinrun's parity tests
#   2 bench line (steady state) of the default build; same-box A/B against the round-5 library and the one-kernel builds of this round
#     (build_opq = hoisted cell lookup + opaque runs, no display list)
#   3 kernel trace + SQ instruction counters of prep / raster
#   4 the whole GPU suite with four workers (without the cffi test, which has its own line)
# usage: bash tools/gpu/r6_call3.sh [tag]
TAG=${1:-r6c3}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PROCGEN_AMD_FATAL_LOG=$R/gpurun_out/${TAG}_fatal.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "display_list or coinrun or render_launch or lds_dma" 2>&1 | tail -12 | tee gpurun_out/${TAG}_pytest_dl.log
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
LIBS=""
for v in build_r05 build_opq build; do [ -f procgen_amd/csrc/$v/libenv.so ] && LIBS=$LIBS,procgen_amd/csrc/$v; done
timeout 600 python tools/gpu/ab_bench.py ${LIBS#,} coinrun 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt -o kt -- python $R/bench.py --steps 64 --warmup 5 --no-cpu-baseline --no-host-landed > $R/gpurun_out/${TAG}_kt.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_kt -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_trace.csv 2>&1
python $R/tests/tools/rocpd_timeline.py $(find /tmp/${TAG}_kt -name "*.db" | head -1) > $R/gpurun_out/${TAG}_timeline.txt 2>&1
head -12 $R/gpurun_out/${TAG}_kernel_trace.csv | cut -c1-180
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace -d /tmp/${TAG}_f0 -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-landed > $R/gpurun_out/${TAG}_f0.log 2>&1
python $R/tests/tools/rocpd_summary.py $(find /tmp/${TAG}_f0 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_f0.csv 2>&1
grep "raster\|4prep\|render_list\|6renderI" $R/gpurun_out/${TAG}_f0.csv | cut -c1-200
cd $R
timeout 900 python -m pytest tests -q -m gpu -n 4 --deselect tests/test_gpu_parity.py::test_boundary_through_cffi 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k cffi 2>&1 | tail -15 | tee gpurun_out/${TAG}_pytest_cffi.log
