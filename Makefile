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

clean:
	rm -rf adaptive-speculative-decoding_amd/lib oracle/_build gpurun_out

.PHONY: build test-cpu test-gpu smoke bench golden clean
