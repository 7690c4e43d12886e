"""Build-container script: reads the parameter descriptors of the ten stock effects out of the reference's sources
(`pub const X: FloatParameter = FloatParameter::new(FourCC(*b"...."), "Name", min..=max, default).with_scaling(..)`,
EnumParameter / BooleanParameter likewise, src/effect/*.rs) and the order in which `Effect::parameters()` lists them, and writes them as DATA to
tests/golden/params.json: ids, names, types, ranges, defaults, scalings, enum variant counts. The .rs text never ships; the fixture does.
tests/test_params_fixture.py holds the whole `pg_effect_kind_param` table of the library (and the oracle's po_params.hpp through the same ABI
shape) against it.

    python tests/golden/make_params.py [/root/reference]
"""
import json
import math
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "src")
FILES = ["gain", "pan", "filter", "eq5", "delay", "reverb", "chorus", "compressor", "gate", "distortion"]   # = pg_effect_kind order


def strip_comments(t):
    return re.sub(r"//[^\n]*", "", t)


def balanced(t, i):
    """t[i] == '(' -> index just behind the matching ')'."""
    depth = 0
    for j in range(i, len(t)):
        if t[j] == "(":
            depth += 1
        elif t[j] == ")":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced")


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def find_enum(name):
    """Variant names of `enum name { .. }` anywhere under src/ (strum VARIANTS = declaration order); `type name = Other;` aliases are followed."""
    for root, _, files in os.walk(SRC):
        for f in files:
            if f.endswith(".rs"):
                m = re.search(r"type\s+" + re.escape(name) + r"\s*=\s*(\w+)\s*;", strip_comments(open(os.path.join(root, f)).read()))
                if m:
                    return find_enum(m.group(1))
    for root, _, files in os.walk(SRC):
        for f in files:
            if not f.endswith(".rs"):
                continue
            t = strip_comments(open(os.path.join(root, f)).read())
            m = re.search(r"enum\s+" + re.escape(name) + r"\s*\{([^}]*)\}", t)
            if m:
                body = re.sub(r"#\[[^\]]*\]", "", m.group(1))
                return [v.split("=")[0].strip() for v in body.split(",") if v.strip()]
    raise KeyError(name)


def consts_of(text, type_name=None):
    """`const NAME: f32 = expr;` of a file (associated consts of any impl in it)."""
    return {m.group(1): m.group(2).strip() for m in re.finditer(r"const\s+([A-Z0-9_]+)\s*:\s*f32\s*=\s*([^;]+);", text)}


def eval_f32(expr, consts):
    e = expr.strip()
    e = re.sub(r"(?:Self|[A-Za-z0-9_]+)::([A-Z][A-Z0-9_]*)", lambda m: "(" + resolve_const(m.group(1), consts) + ")", e)
    e = e.replace("PI as f32", "F32PI").replace("as f32", "")
    e = re.sub(r"\bPI\b", "math.pi", e)
    e = e.replace("F32PI", "float(np.float32(math.pi))")
    return float(np.float32(eval(e, {"math": math, "np": np})))


def resolve_const(name, consts):
    if name in consts:
        return consts[name]
    for root, _, files in os.walk(SRC):      # e.g. DistortionType::MAX_DRIVE lives in the same file; fall back to a tree search
        for f in files:
            if f.endswith(".rs"):
                c = consts_of(strip_comments(open(os.path.join(root, f)).read()))
                if name in c:
                    return c[name]
    raise KeyError(name)


def parse_effect(fname):
    text = strip_comments(open(os.path.join(SRC, "effect", fname + ".rs")).read())
    consts = consts_of(text)
    effect_name = re.search(r'EFFECT_NAME\s*:\s*&str\s*=\s*"([^"]+)"', text).group(1)
    # every `XParameter::new(..)` with its builder calls, keyed by the const it initialises (arrays: NAME[i])
    decls = {}
    for m in re.finditer(r"pub const\s+([A-Z0-9_]+)\s*:\s*(\[?)\s*(Float|Enum|Boolean|Integer)Parameter", text):
        const, is_array = m.group(1), m.group(2) == "["
        end = text.index(";\n", text.index("=", m.end()))        # (array types hold a `;` of their own: start behind the `=`)
        body = text[m.end():end]
        idx = 0
        for n in re.finditer(r"(Float|Enum|Boolean|Integer)Parameter::new\s*\(", body):
            a0 = n.end() - 1
            a1 = balanced(body, a0)
            args = split_args(body[a0 + 1:a1 - 1])
            tail = body[a1:]
            nxt = re.search(r"(Float|Enum|Boolean|Integer)Parameter::new\s*\(", tail)
            tail = tail[:nxt.start()] if nxt else tail
            kind = n.group(1)
            d = {"id": re.search(r'b"(.{4})"', args[0]).group(1), "name": args[1].strip('"'), "type": {"Float": "float", "Enum": "enum", "Boolean": "bool", "Integer": "int"}[kind],
                 "scaling": "linear", "scaling_args": [], "n_values": 0}
            if kind == "Float":
                lo, hi = args[2].split("..=")
                d["min"], d["max"], d["default"] = eval_f32(lo, consts), eval_f32(hi, consts), eval_f32(args[3], consts)
                sc = re.search(r"\.with_scaling\(\s*ParameterScaling::(\w+)\(([^)]*)\)", tail)
                if sc:
                    d["scaling"] = sc.group(1).lower()
                    d["scaling_args"] = [eval_f32(a, consts) for a in split_args(sc.group(2))]
            elif kind == "Enum":
                variants = find_enum(re.match(r"(\w+)::VARIANTS", args[2]).group(1))
                dm = re.match(r"(\w+)::(\w+)\s+as\s+usize", args[3])
                default = variants.index(dm.group(2)) if dm else int(args[3])
                d["min"], d["max"], d["default"], d["n_values"] = 0.0, float(len(variants) - 1), float(default), len(variants)
                d["variants"] = variants
            elif kind == "Boolean":
                d["min"], d["max"], d["default"] = 0.0, 1.0, 1.0 if args[2].strip() == "true" else 0.0
            else:
                raise NotImplementedError(kind)
            decls[f"{const}[{idx}]" if is_array else const] = d
            idx += 1
    # struct fields -> consts (new(): `field: X::from_description(Self::CONST)` / `field: Self::CONSTS.map(`)
    field_const = {}
    for m in re.finditer(r"(\w+)\s*:\s*(?:\w+::)*from_description\(\s*Self::([A-Z0-9_]+)\s*\)", text):
        field_const.setdefault(m.group(1), m.group(2))
    for m in re.finditer(r"(\w+)\s*:\s*Self::([A-Z0-9_]+)\s*\.map\(", text):
        field_const.setdefault(m.group(1), m.group(2))
    for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)\s*=\s*(?:\w+::)*from_description\(\s*Self::([A-Z0-9_]+)\s*\)", text):   # `let field = ..; Self { field, .. }`
        field_const.setdefault(m.group(1), m.group(2))
    body = re.search(r"fn parameters\(&self\)[^{]*\{(.*?)\n    \}", text, re.S).group(1)
    order = []
    for m in re.finditer(r"self\.(\w+)(?:\[(\d+)\])?\s*\.description\(\)", body):
        const = field_const[m.group(1)]
        order.append(decls[f"{const}[{m.group(2)}]" if m.group(2) is not None else const])
    assert len(order) == len(decls), (fname, len(order), len(decls))
    w = re.search(r"fn weight\(&self\)\s*->\s*usize\s*\{\s*(\d+)", text)
    return {"name": effect_name, "weight": int(w.group(1)) if w else None, "parameters": order}


if __name__ == "__main__":
    out = {"source": "emuell/phonic v0.16.0 src/effect/*.rs (parameter consts + Effect::parameters() order)", "effects": [parse_effect(f) for f in FILES]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "params.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, sum(len(e["parameters"]) for e in out["effects"]), "parameters")
