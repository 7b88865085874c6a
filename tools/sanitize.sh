#!/bin/bash
# Host-side checker target (SURVEY section 5 "Race detection / sanitizers": the reference has none, core/CMakeLists.txt:235).
#   tools/sanitize.sh            builds lib/libsnn_core_asan.so (-fsanitize=address,undefined) and runs the CPU suite's host tests against it
#   tools/sanitize.sh gpu        on a GPU box: the smoke test + the host-mirror GPU tests under the same library, plus the kernels under
#                                AMD_LOG_LEVEL=1 HSA_ENABLE_DEBUG=1 (out-of-bounds accesses of a kernel show up as a memory-access fault that aborts the run)
#   tools/sanitize.sh guard      on a GPU box: the device-side guard mode (SNNHIP_GUARD=1, include/snnhip.h: red zones of 0xFF around every device allocation,
#                                checked at every snnhip_sync / download) over the randomised convolution / operator sweeps, the inverted-residual kernels,
#                                the fp16 wide / marching kernels, the K-split and Winograd kernels and the bench-size configs c1-c4, plus the deliberately broken build the checker must catch
set -eu
cd "$(dirname "$0")/.."
if [ "${1:-cpu}" = "guard" ]; then
  python -c "import __graft_entry__ as g; g.build_hip(); g.build_host()"
  [ -f build/abl/libsnnhip_guardbreak.so ] || tools/exp_one.sh conv2d_generic.hip guardbreak:-DSNNHIP_GUARD_BREAK=1 > /dev/null
  python -m pytest tests/test_guard_gpu.py -q -m gpu -x
  SNNHIP_GUARD=1 python -m pytest tests/test_conv_fuzz_gpu.py tests/test_ops_fuzz_gpu.py tests/test_irb_gpu.py tests/test_conv_wide_gpu.py tests/test_conv_widep_gpu.py tests/test_conv_ksplit_gpu.py tests/test_conv_wino_gpu.py tests/test_espcn_gpu.py -q -m gpu -x
  SNNHIP_GUARD=1 python -m pytest tests/test_configs_gpu.py tests/test_golden.py -q -m gpu -x -k "not c5"
  exit 0
fi
python -c "import __graft_entry__ as g; g.build_hip(); g.build_host(); print(g.build_host(sanitize='address,undefined'))"
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
export LD_PRELOAD="$ASAN:$UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export SNN_CORE_LIB_PATH="$PWD/shadernn_amd/lib/libsnn_core_asan.so"
if [ "${1:-cpu}" = "gpu" ]; then
  AMD_LOG_LEVEL=1 HSA_ENABLE_DEBUG=1 python -m pytest tests/test_host.py tests/test_configs_gpu.py -q -m gpu -x -k "not c5"
else
  python -m pytest tests/test_host.py tests/test_param_import.py -q -m "not gpu" -x
fi
