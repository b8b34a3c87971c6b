import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def relerr(a, b):
    """max|a-b| / max|b|  (values are clamp-bounded, avoid blow-up near zeros)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle as orc
    orc.lib()
    return orc


def digest_errors(named, g, prefix, exclude=()):
    """Compare an ordered {name: array} map with a digest written by tests/golden/make_golden.py:grad_digest.
    Returns the worst errors, each relative to the tensor's golden L2 norm (floored at 1e-2 of the largest tensor norm:
    the gradient of a conv bias that feeds a GroupNorm is analytically zero and numerically round-off noise):
      norm  |‖a‖-‖g‖| / ‖g‖ ;  proj  |<a,r>-<g,r>| / ‖g‖ (r ~ N(0,1): an element-sensitive checksum) ;
      head  max|a[:8]-g[:8]| / (‖g‖/sqrt(n)) ;  full  max|a-g| / max|g| and  full_l2  ‖a-g‖/‖g‖ for the tensors stored whole."""
    from sin3dm_amd import testing as T
    worst = {"norm": 0.0, "proj": 0.0, "head": 0.0, "full": 0.0, "full_l2": 0.0}
    gn, gp, gh = g[f"{prefix}/norm"], g[f"{prefix}/proj"], g[f"{prefix}/head"]
    names = [str(k) for k in g[f"{prefix}/names"]]
    assert sorted(names) == sorted(named)
    for i, k in enumerate(names):
        if k in exclude:
            continue
        a = np.asarray(named[k], dtype=np.float64).reshape(-1)
        n = max(float(gn[i]), 1e-2 * float(np.max(gn)), 1e-30)
        r = T.synthetic_tensor("digest/" + k, (a.size,), 7).astype(np.float64)
        worst["norm"] = max(worst["norm"], abs(float(np.linalg.norm(a)) - gn[i]) / n)
        worst["proj"] = max(worst["proj"], abs(float(a @ r) - gp[i]) / n)
        m = min(8, a.size)
        worst["head"] = max(worst["head"], float(np.max(np.abs(a[:m] - gh[i][:m]))) / (n / np.sqrt(a.size)))
        key = f"{prefix}/full/{k}"
        if key in g.files:
            worst["full_l2"] = max(worst["full_l2"], float(np.linalg.norm(a - g[key].reshape(-1))) / n)
            worst["full"] = max(worst["full"], float(np.max(np.abs(a - g[key].reshape(-1)))) /
                                max(float(np.max(np.abs(g[key]))), 1e-2 * float(np.max(gn)) / np.sqrt(a.size)))
    return worst


def zero_grad_params(tag="mc32_a", file="train_grads"):
    """Parameters whose gradient is analytically zero (a conv bias feeding a Group/InstanceNorm): what AdamW does with
    their round-off-noise gradients is not comparable between implementations."""
    g = golden(file)
    prefix = f"{tag}.grad" if tag else "grad"
    n, names = g[f"{prefix}/norm"], g[f"{prefix}/names"]
    return {str(k) for k, v in zip(names, n) if v < 1e-6 * float(np.max(n))}
