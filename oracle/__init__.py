"""CPU oracle for the LMIP hot path — TEST INFRASTRUCTURE, never imported by the product.

``lmip_oracle.c`` / ``lmip.py``: restatement of the reference's WGSL shaders (render parity unpinned).
``ring_oracle.py``: restatement of the reference's ring-buffer host logic (pinned by the reference's tests).
"""
