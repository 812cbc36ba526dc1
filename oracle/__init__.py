"""CPU oracle for the LDMSeg denoising path.  TEST INFRASTRUCTURE ONLY.

This package is a plain torch-CPU / numpy restatement of the reference's
algorithm for the hot path (scheduler, seg-VAE, UNet forward, sampling loop).
It exists to CHECK the HIP product path; nothing under
``latent-diffusion-segmentation_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

Pinning status (see DESIGN.md "Oracle"):
  * ddim.py  - pinned: bit-exact against goldens produced by importing
               /root/reference/ldmseg/schedulers/ddim_scheduler.py
               (tests/golden/make_golden.py -> tests/golden/scheduler.npz).
  * vae.py   - pinned: against goldens produced by importing
               /root/reference/ldmseg/models/vae.py::GeneralVAESeg
               (tests/golden/vae.npz).
  * unet.py  - PARITY UNPINNED by the reference: the block arithmetic lives in
               the third-party package diffusers==0.16.1 (data/environment.yml:50)
               which is neither vendored under /root/reference nor installed
               here.  Restated from its published architecture; pinned only
               structurally (param count 859,520,964 / 815,556,484, tensor
               count 686 / 574, key set, shapes) - SURVEY.md Appendix A.
  * sample.py- pinned: loop semantics checked by running the *reference
               scheduler object* inside the same loop (tests/golden/sample_loop.npz).
"""
