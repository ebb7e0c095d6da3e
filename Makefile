# Convenience targets; the real build lives in effectivediffusivityfvm_amd/csrc/Makefile and oracle/Makefile.
PY ?= python

build:            ## libdeff_amd.so + deff2d (hipcc, gfx950) and the CPU oracle (tests only)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## CPU suite (oracle vs golden vectors, host logic, ABI symbols, front-end fuzz under ASan/UBSan)
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## parity suite, needs an MI355X
	$(PY) -m pytest tests -q -m gpu

bench:            ## headline benchmark, one JSON line
	$(PY) bench.py

clean:
	$(MAKE) -C effectivediffusivityfvm_amd/csrc clean
	$(MAKE) -C oracle clean
	$(MAKE) -C tests/cpp clean

.PHONY: build test test-gpu bench clean
