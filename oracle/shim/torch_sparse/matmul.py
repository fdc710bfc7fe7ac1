"""submodule placeholder: `import torch_sparse.matmul as matmul` (reference utils.py:6) resolves to the package attribute, which
torch_sparse/__init__ (and this shim's) rebinds to the function"""
