def create_approx_func(*a, **k):
    raise NotImplementedError("mitlef is not available; Cole-Cole/zga bases are out of scope")


def ml_pade_approx(*a, **k):
    raise NotImplementedError("mitlef is not available; Cole-Cole/zga bases are out of scope")
