# Convenience targets (the driver uses __graft_entry__.py / bench.py / pytest directly).
PY ?= python

build:            ## hipcc --offload-arch=gfx950 -> adaptive-speculative-decoding_amd/lib/libasd_hip.so, gcc -> oracle
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu:         ## oracle vs goldens, host logic, ABI symbols, gloo 2-rank exchange (no GPU needed)
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## parity of the HIP path through the C ABI (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu

smoke:
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench:
	$(PY) bench.py

golden:           ## regenerate tests/golden from the reference's own files (dev container only)
	PYTHONDONTWRITEBYTECODE=1 $(PY) oracle/gen_golden.py

test-hooks:       ## the TEST build of the library (libasd_hip_test.so: product objects + the asd_debug_* switches under -DASD_TEST_HOOKS)
	$(PY) adaptive-speculative-decoding_amd/build.py --test-hooks

asan-host:        ## host side of the launchers under AddressSanitizer + UBSan (CPU only: the ABI / argument-check tests)
	$(PY) adaptive-speculative-decoding_amd/build.py --asan
	ASD_LIB_PATH=$(CURDIR)/adaptive-speculative-decoding_amd/lib/libasd_hip_asan.so \
	LD_PRELOAD=$$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) \
	ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0:verify_asan_link_order=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	$(PY) -m pytest tests/test_abi.py -q -p no:cacheprovider

clean:
	rm -rf adaptive-speculative-decoding_amd/lib oracle/_build gpurun_out

.PHONY: build test-cpu test-gpu smoke bench golden test-hooks asan-host clean
