"""The ctypes stub INTEGRATION.md hands to a maintainer of the reference must describe the SAME structs as include/mtts.h
(bound in meta_tts_amd/_lib.py): a missing trailing field hands mtts_set_batches stack garbage.  The two ctypes.Structure
blocks of the stub are executed as written and compared field by field with the binding's."""
import ctypes as C
import os
import re

from meta_tts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_structs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "class mtts_batch(C.Structure)" in b)
    out = {}
    for name in ("mtts_model_cfg", "mtts_batch"):
        m = re.search(r"(class %s\(C\.Structure\):.*?)\n(?=\S)" % name, code, flags=re.S)
        assert m, f"stub has no ctypes.Structure for {name}"
        ns = {"C": C}
        exec(m.group(1), ns)
        out[name] = ns[name]
    return out


def _layout(st):
    return [(n, getattr(st, n).offset, getattr(st, n).size) for n, _ in st._fields_]


def test_stub_structs_match_binding():
    stub = _stub_structs()
    for name, ref in (("mtts_model_cfg", _lib.ModelCfg), ("mtts_batch", _lib.Batch)):
        assert C.sizeof(stub[name]) == C.sizeof(ref), name
        assert _layout(stub[name]) == _layout(ref), name


def test_header_structs_list_the_same_fields():
    """include/mtts.h is the source of truth: every member of its two structs appears, in order, in the binding."""
    hdr = open(os.path.join(ROOT, "include", "mtts.h")).read()
    for cname, ref in (("mtts_model_cfg", _lib.ModelCfg), ("mtts_batch", _lib.Batch)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(int64_t|int|float|unsigned)\s*\*?", "", decl)
            names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
        assert names == [n for n, _ in ref._fields_], cname
