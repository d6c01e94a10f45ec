# Convenience targets; the driver's entry points are __graft_entry__.build() / smoke() and bench.py.
.PHONY: all lib oracle test-cpu clean
all: lib oracle
lib:            ## libfmmbem_hip.so for gfx950 (hipcc cross-compiles without a GPU)
	$(MAKE) -C fmm-bem-relaxed_amd/csrc -j4
oracle:         ## CPU restatement of the reference path -- test infrastructure only
	$(MAKE) -C oracle
test-cpu: all   ## what runs without a GPU
	python -m pytest tests -x -q -m "not gpu"
clean:
	$(MAKE) -C fmm-bem-relaxed_amd/csrc clean
	$(MAKE) -C oracle clean
