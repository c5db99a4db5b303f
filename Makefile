# Convenience targets; the driver uses __graft_entry__.build(), pytest and bench.py directly.
.PHONY: build test-cpu test-gpu bench sweep clean

build:
	python -c "import __graft_entry__ as g; g.build()"

test-cpu: build
	python -m pytest tests -x -q -m "not gpu"

test-gpu: build          # needs an MI355X
	python -m pytest tests -x -q -m gpu

bench: build             # needs an MI355X
	python bench.py

sweep: build             # needs an MI355X
	python tools/sweep.py --cases quick

clean:
	$(MAKE) -C fastlanes_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -f examples/*.so tests/cpp/test_trait_mirror tools/abbench
