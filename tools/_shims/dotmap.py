"""Throw-away stand-in for the `dotmap` package (not installed here): attribute access,
auto-create on miss, toDict().  Harness code for tools/gen_golden.py only."""


class DotMap(dict):
    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        if k not in self:
            self[k] = DotMap()
        return self[k]

    def __setattr__(self, k, v):
        self[k] = v

    def toDict(self):
        return {k: (v.toDict() if isinstance(v, DotMap) else v) for k, v in self.items()}
