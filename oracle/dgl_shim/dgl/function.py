"""TEST INFRASTRUCTURE ONLY -- restatement of dgl.function's two opaque descriptors used by the
reference (learner.py:9,38-39,44-45).  See oracle/dgl_shim/dgl/__init__.py."""


def copy_src(src, out):
    return ('copy_src', src, out)


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return ('sum', msg, out)
