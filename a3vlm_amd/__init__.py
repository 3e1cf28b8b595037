"""a3vlm_amd -- MI355X (gfx950) native hot path of A3VLM.

csrc/            hand-written HIP kernels + the C-ABI (include/a3vlm_hip.h)
lib.py / ops.py  ctypes loader and thin tensor wrappers (PyTorch only owns memory/streams)
model/           host-side mirror of the reference plugin interface
                 (accessory.model.LLM.llama_ens5 / accessory.model.meta)
"""
__version__ = "0.1.0"
