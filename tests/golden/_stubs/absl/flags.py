"""Dict-backed FLAGS with DEFINE_* helpers (just enough for the reference's module-level flag
definitions). Test-fixture tooling only."""
class _Flags(object):
    def __init__(self):
        object.__setattr__(self, '_v', {})
    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, '_v')[k]
        except KeyError:
            raise AttributeError(k)
    def __setattr__(self, k, v):
        self._v[k] = v
    def __call__(self, argv):
        return argv
FLAGS = _Flags()
def _define(name, default, help=None, **kw):
    if name not in FLAGS._v:
        FLAGS._v[name] = default
DEFINE_integer = DEFINE_float = DEFINE_string = DEFINE_boolean = DEFINE_bool = DEFINE_list = _define
