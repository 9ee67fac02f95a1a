"""Extract the API surface the reference's own test-suite drives: every `MultiModalMuSig.<name>(...)` call in
/root/reference/test/{lda,ilda,mmctm,immctm,common}.jl with the number of positional arguments and keyword names at each call site,
plus every `model.<field>` the tests read or assign.  Output: tests/golden/reference_test_api.json -- a NAME LIST (no reference text).
Run in the build container (the reference does not travel): python tests/golden/make_test_api.py"""
import json
import os
import re

REF = "/root/reference/test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_test_api.json")
FILES = ["lda.jl", "ilda.jl", "mmctm.jl", "immctm.jl", "common.jl"]
IDENT = r"[^\W\d][\w!]*"       # Julia identifier (unicode letters, digits, !)


def strip_comments(txt):
    return "\n".join(re.sub(r"#.*$", "", ln) for ln in txt.split("\n"))


def balanced(txt, i):
    depth = 0
    for j in range(i, len(txt)):
        if txt[j] in "([{":
            depth += 1
        elif txt[j] in ")]}":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced")


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch in ",;" and depth == 0:
            out.append((cur.strip(), ch)); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append((cur.strip(), ""))
    return out


def main():
    calls, fields = {}, {}
    for fn in FILES:
        txt = strip_comments(open(os.path.join(REF, fn), encoding="utf-8").read())
        for m in re.finditer(r"MultiModalMuSig\.(" + IDENT + r")\s*\(", txt):
            name = m.group(1)
            end = balanced(txt, m.end() - 1)
            parts = split_top(txt[m.end():end - 1])
            npos, kws, after_semicolon = 0, [], False
            for arg, sep in parts:
                km = re.match(r"^(" + IDENT + r")\s*=(?!=)", arg)
                if after_semicolon or km:
                    kws.append(km.group(1) if km else arg)
                else:
                    npos += 1
                if sep == ";":
                    after_semicolon = True
            line = txt.count("\n", 0, m.start()) + 1
            sig = {"npos": npos, "kw": sorted(kws)}
            e = calls.setdefault(name, {"sites": [], "signatures": []})
            e["sites"].append("test/%s:%d" % (fn, line))
            if sig not in e["signatures"]:
                e["signatures"].append(sig)
        model_kind = {"lda.jl": "LDA", "ilda.jl": "ILDA", "mmctm.jl": "MMCTM", "immctm.jl": "IMMCTM", "common.jl": "MMCTM"}[fn]
        for m in re.finditer(r"\b(?:model|newmodel)\.(" + IDENT + r")", txt):
            fields.setdefault(model_kind, set()).add(m.group(1))
    out = {"source": "names used by /root/reference/test/{lda,ilda,mmctm,immctm,common}.jl (comments stripped; commented-out tests are not included)",
           "calls": {k: calls[k] for k in sorted(calls)},
           "fields": {k: sorted(v) for k, v in sorted(fields.items())}}
    with open(OUT, "w", encoding="utf-8") as fh:
        json.dump(out, fh, ensure_ascii=False, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote %s: %d functions, fields of %d model types" % (OUT, len(calls), len(fields)))


if __name__ == "__main__":
    main()
