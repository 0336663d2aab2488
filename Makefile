# Convenience targets; the driver's contract lives in __graft_entry__.py and bench.py.
.PHONY: build test test-gpu bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	$(MAKE) -C vistrace_amd/csrc clean
	$(MAKE) -C oracle clean
	$(MAKE) -C tests/cpp clean
