"""CPU oracle of the lit-parrot quantized decode path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the reference path (generate/base.py::generate ->
lit_gpt/model.py::GPT.forward -> quantize/{gptq,bnb}.py Linears) so that the HIP kernels can be checked on a GPU
box where the reference itself is absent.  It is pinned to the reference by ``tests/golden/*`` (vectors produced by
running the reference in the build container, see tests/golden/make_golden.py) — except the LLM.int8 arithmetic,
which lives in the third-party bitsandbytes wheel (>= 0.40.0, unpinned, absent here): that part is **parity
unpinned** and restates the published algorithm.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``lit-parrot_amd/`` does, and the product has no CPU fallback.
"""
