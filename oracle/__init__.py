"""CPU oracle for the DFoT denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 restatement (CPU) of the algorithm the
reference (ktncktnc/diffusion-forcing-transformer) runs on its sampling path:
schedule buffers, camera-ray encoding, the UViT3DPose backbone, the DDIM /
v-prediction step, History-Guidance prepare/compose and the sampler drivers.
Every function cites the reference file:line it restates.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import this package -- and only as the checker;
  * the product package (``diffusion-forcing-transformer_amd``) never imports
    it and has no CPU fallback: it fails loudly when the HIP library is absent.

Parity pins: ``tests/golden/*.npz`` were produced by ``tools/make_golden.py``
which executes the reference's OWN source files on CPU in the build container
(third-party packages the image lacks -- timm / diffusers / rotary_embedding_torch
/ omegaconf / lightning -- replaced by minimal stand-ins restating the pinned
upstream semantics, see that script's header).  ``tests/test_oracle_golden.py``
checks this oracle against every one of those vectors.  The reference itself
ships no tests or golden vectors for this path, so the
stand-in semantics of those third-party layers are "parity unpinned by the
reference"; everything else is pinned by the reference's own code as executed.
"""
