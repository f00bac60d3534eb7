"""numba is absent in the build container; jit becomes the identity decorator, so the reference's
time_warp body runs as plain Python on numpy f32 scalars (numerically identical). Fixture tooling only."""
def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
