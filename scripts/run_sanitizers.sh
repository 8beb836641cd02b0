#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything that runs on the CPU (the reference has an ASAN CI
# job: CMakeLists.txt:383-404, .github/workflows/main.yml:39-44).  GPU AddressSanitizer is not available on the
# pool, so device code is not covered:
#   1. the oracle (oracle/libhdk_oracle_asan.so) under the whole CPU test-suite;
#   2. the HOST half of libhdk_hip.so (plan validation, matchers, shape / workspace arithmetic, the C ABI's argument
#      checks; the kernel-routing matchers through tests/test_kernel_routing_cpu.py) -- built with -fsanitize=address,undefined for the host pass only (-fno-gpu-sanitize) as
#      hdk_amd/libhdk_hip_asan.so -- under the ABI tests that need no device;
#   3. the C++ binding harness's host code (tests/cpp), compiled with the same flags (compile + link; it needs a
#      device to run).
# Usage: bash scripts/run_sanitizers.sh [pytest args]      exit status 0 = no report
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
ASAN_RT=$(gcc -print-file-name=libasan.so)
UBSAN_RT=$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
echo "== 1. oracle under ASAN/UBSan: CPU test-suite"
make -C oracle --no-print-directory asan
LD_PRELOAD="$ASAN_RT $UBSAN_RT" HDK_ORACLE_LIB=$ROOT/oracle/libhdk_oracle_asan.so \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
echo "== 2. host half of libhdk_hip.so under ASAN/UBSan: ABI tests"
make -C hdk_amd/csrc variant NAME=asan DEFS="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-gpu-sanitize" \
  LDEXTRA="-fsanitize=address,undefined -fno-gpu-sanitize" -j4 > /tmp/asan_build.log 2>&1 || { tail -20 /tmp/asan_build.log; exit 1; }
CLANG_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true)
LD_PRELOAD="${CLANG_RT:-$ASAN_RT}" HDK_HIP_LIB=$ROOT/hdk_amd/libhdk_hip_asan.so \
  python -m pytest tests/test_abi.py tests/test_abi_negative.py tests/test_plan_layout.py tests/test_kernel_routing_cpu.py -x -q -p no:cacheprovider
echo "== 3. C++ binding harness, host code with the same flags (compile + link only)"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Wall -Werror -Wno-unused-parameter -I /root/reference/omniscidb -I include \
  -I hdk_amd/glue -I tests/cpp tests/cpp/harness.cpp -L hdk_amd -lhdk_hip -Wl,-rpath,"$ROOT/hdk_amd" -o /tmp/harness_asan 2>&1 | tail -20
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Wall -Werror -Wno-unused-parameter -D__HIP_PLATFORM_AMD__ -I /root/reference/omniscidb -I include \
  -I hdk_amd/glue -I tests/cpp -I /opt/rocm/include tests/cpp/multi_device.cpp -L hdk_amd -lhdk_hip -L /opt/rocm/lib -lrccl -lamdhip64 -lpthread \
  -Wl,-rpath,"$ROOT/hdk_amd" -Wl,-rpath,/opt/rocm/lib -o /tmp/multi_device_asan 2>&1 | tail -20
# (pipefail: a failed compile fails the script -- it used to read as a pass)
echo "sanitizers: no report"
