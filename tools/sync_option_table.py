#!/usr/bin/env python3
"""Rewrites the fenced block of INTEGRATION.md section 5 with the text bang_describe_options() prints (tests/test_cabi.py checks it)."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
import bang_amd  # noqa: E402

bang_amd.build()
lib = C.CDLL(bang_amd.binding.lib_path())
lib.bang_describe_options.argtypes = [C.c_char_p, C.c_size_t]
need = lib.bang_describe_options(None, 0)
buf = C.create_string_buffer(need)
lib.bang_describe_options(buf, need)
p = os.path.join(ROOT, "INTEGRATION.md")
doc = open(p).read()
new, n = re.subn(r"```\noptions \(bang_set_option key.*?```", lambda m_: "```\n" + buf.value.decode().rstrip() + "\n```", doc, count=1, flags=re.S)
assert n == 1
open(p, "w").write(new)
print("INTEGRATION.md section 5 updated" if new != doc else "INTEGRATION.md section 5 already current")
