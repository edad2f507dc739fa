"""`dgl.utils.expand_as_pair` counterpart (reference src/no-sampling/models.py:12, :350, :420)."""


def expand_as_pair(input_, g=None):
    if isinstance(input_, tuple):
        return input_
    if g is not None and getattr(g, "is_block", False):
        raise NotImplementedError("blocks belong to the sampled scripts, outside the full-batch path")
    return input_, input_
