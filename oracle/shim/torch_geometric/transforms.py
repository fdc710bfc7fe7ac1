from .utils import to_undirected


class ToUndirected:
    """In-place semantics (PyG <= 2.2): the reference discards the return value
    (main_graph_knowledge_transfer.py:410-411)."""
    def __init__(self, reduce="add", merge=True):
        self.merge = merge

    def __call__(self, data):
        data.edge_index = to_undirected(data.edge_index, num_nodes=data.num_nodes)
        return data
