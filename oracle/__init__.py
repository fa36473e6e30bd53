"""CPU oracle for the MV-VDM denoising UNet forward (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``animate3d_amd``) never does.
"""
