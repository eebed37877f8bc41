"""CPU oracle for the latent-video denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported CPU baseline.
The product path (``latent-flexible-video-diffusion-modeling_amd/``) never
imports this package and fails loudly when its HIP library is missing.

The oracle is a functional fp32 restatement (PyTorch CPU ops on plain tensors,
numpy float64 for the diffusion tables) of the reference algorithm in
``improved_diffusion/{unet,rpe,nn,gaussian_diffusion,respace}.py``.  Parity is
PINNED: ``oracle/make_golden.py`` imports the real reference from
``/root/reference`` (build container only), runs it on seeded inputs with the
closed-form parameter recipe of ``oracle/recipe.py`` and stores inputs+outputs
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this oracle
against those vectors on every run.
"""
