def matmul(*a, **k):
    raise NotImplementedError("oracle shim placeholder")
