"""Test helper: the ctypes stub INTEGRATION.md §3 shows a maintainer of the reference (vietTTS/hifigan/mel2wave.py:20-41),
extracted from the document so that document and ABI cannot drift apart."""
import re
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def stub_source() -> str:
    """The §3 code block, verbatim."""
    md = (REPO / "INTEGRATION.md").read_text()
    sec = md[md.index("## 3. The stub a maintainer would add") :]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md §3 lost its python block"
    return m.group(1)


def runnable_source(lib_path: str, ckpt_dir: str) -> str:
    """The block with exactly two lines adapted to run outside the reference's package: the reference's own
    `from .config import FLAGS` (vietTTS/hifigan/config.py:5-6: FLAGS.ckpt_dir) becomes a local stand-in, and the library path
    (relative to the reference checkout in the document) becomes absolute."""
    src = stub_source()
    assert src.count("from .config import FLAGS") == 1 and src.count('"viettts_amd/lib/libvtts_hifigan.so"') == 1
    src = src.replace("from .config import FLAGS",
                      f"from pathlib import Path as _P\nclass FLAGS:\n    ckpt_dir = _P({ckpt_dir!r})")
    return src.replace('"viettts_amd/lib/libvtts_hifigan.so"', repr(lib_path))
