import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
# repo root (oracle/ as a package-less module dir) and the product package dir
for p in (ROOT, ROOT / "kompass-core_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
