"""pytest configuration: registers the `gpu` marker and makes the repo root, the package directory
(`eddie-wang-hackathon2023_amd/`, a script directory laid out like the reference's
examples/whisper/) and the oracle importable."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "eddie-wang-hackathon2023_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import native  # noqa: E402,F401  (sets the package's HIP-runtime defaults before any test initialises the runtime: the suite runs what ships)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_BUILD_ERROR = None


def pytest_sessionstart(session):
    """The C-ABI library is built in-tree (git-ignored).  `make` runs on every session: a timestamp no-op when the
    library is current, a rebuild after any edit of csrc/ (a stale .so would let the GPU suite pass for kernels
    that no longer exist).  hipcc cross-compiles gfx950 without a GPU, exactly as __graft_entry__.build() does.
    Without hipcc only the tests that need the library are skipped -- the oracle and host-logic tests still run.
    With hipcc present a FAILED build ends the session with an error: skipping would report a green run for kernels
    that do not compile."""
    global _BUILD_ERROR
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        if not os.path.exists(os.path.join(PKG, "libwhisper_mi355.so")):
            _BUILD_ERROR = "hipcc not found and libwhisper_mi355.so is not built"
        return
    r = subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        pytest.exit("building libwhisper_mi355.so failed:\n" + r.stdout[-3000:], returncode=2)


def pytest_collection_modifyitems(config, items):
    if _BUILD_ERROR is None:
        return
    skip = pytest.mark.skip(reason=_BUILD_ERROR)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
