"""Static check of the Julia shim (multimodalmusig.jl_amd/julia/MultiModalMuSigHIP.jl) against include/mmmusig.h.

Julia is absent from the build container and the GPU boxes, so the shim cannot be executed; what can be verified is that every
`ccall((:mmm_..., LIB), Ret, (ArgTypes...), args...)` names a function the header declares, with the same number of
arguments, the same C type for each (Cint <-> int, Cdouble <-> double, Csize_t <-> size_t, Ptr/Ref{T} <-> T*, Cstring <->
const char*), the same return type, and as many values as types.  The reference side of the boundary these calls replace:
MultiModalMuSig.jl:9 (exports), LDA.jl:198, MMCTM.jl:457-458, IMMCTM.jl:437 (fit!)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "multimodalmusig.jl_amd", "julia", "MultiModalMuSigHIP.jl")
HDR = os.path.join(ROOT, "include", "mmmusig.h")


def _split_top(s):
    """split at top-level commas"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(text, i):
    """text[i] == '(' -> index just past its matching ')'"""
    depth = 0
    for j in range(i, len(text)):
        if text[j] == "(":
            depth += 1
        elif text[j] == ")":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced")


def header_prototypes():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    protos = {}
    for m in re.finditer(r"\b((?:const\s+)?[A-Za-z_][A-Za-z_0-9]*\s*\**)\s*\b(mmm_[a-z_A-Z0-9]+)\s*\(", txt):
        ret, name = m.group(1).strip(), m.group(2)
        end = _balanced(txt, m.end() - 1)
        if not txt[end:].lstrip().startswith(";"):
            continue
        params = txt[m.end():end - 1].strip()
        args = [] if params in ("", "void") else _split_top(params)
        protos[name] = (ctype(ret), [ctype(a) for a in args])
    return protos


def ctype(decl):
    """C declaration -> category"""
    d = decl.strip()
    if "*" in d or "[" in d:
        base = re.sub(r"\bconst\b", "", d.split("*")[0].split("[")[0]).split()
        base = base[0] if base else "void"
        if d.count("*") >= 2:
            return "ptr:ptr"
        if "[" in d:                # `double terms[7]`, `char out[64]`, `int out[8]`
            base = re.sub(r"\bconst\b", "", d.split("[")[0]).split()[0]
        return "ptr:" + {"mmm_ctx": "void", "mmm_lda": "void", "mmm_ctm": "void", "void": "void", "char": "char", "double": "double", "int": "int",
                         "int32_t": "int32", "int64_t": "int64", "mmm_solver_opts": "void", "mmm_tuning_opts": "void", "size_t": "size_t"}.get(base, base)
    toks = re.sub(r"\bconst\b", "", d).split()
    t = toks[0]
    return {"int": "int", "double": "double", "size_t": "size_t", "void": "void", "int64_t": "int64"}.get(t, t)


def jtype(t):
    t = t.strip()
    simple = {"Cint": "int", "Cdouble": "double", "Csize_t": "size_t", "Cstring": "ptr:char", "Cvoid": "void", "Int32": "int32", "Int64": "int64"}
    if t in simple:
        return simple[t]
    m = re.fullmatch(r"(Ptr|Ref)\{(.+)\}", t)
    assert m, "unknown Julia C type %r" % t
    inner = m.group(2).strip()
    if inner.startswith(("Ptr{", "Ref{")):
        return "ptr:ptr"
    return "ptr:" + {"Cvoid": "void", "Cdouble": "double", "Cint": "int", "Int32": "int32", "Int64": "int64", "UInt8": "char", "Cchar": "char",
                     "SolverOpts": "void", "TuningOpts": "void"}.get(inner, inner)


def shim_ccalls():
    txt = open(SHIM).read()
    txt = re.sub(r"#[^\n]*", "", txt)
    calls = []
    for m in re.finditer(r"ccall\s*\(", txt):
        end = _balanced(txt, m.end() - 1)
        parts = _split_top(txt[m.end():end - 1])
        fm = re.fullmatch(r"\(\s*:(mmm_[a-z_A-Z0-9]+)\s*,\s*LIB\s*\)", parts[0])
        assert fm, "ccall target not of the form (:mmm_x, LIB): %r" % parts[0]
        name, ret = fm.group(1), jtype(parts[1])
        tt = parts[2].strip()
        assert tt.startswith("(") and tt.endswith(")")
        types = [jtype(x) for x in _split_top(tt[1:-1]) if x.strip()]
        calls.append((name, ret, types, len(parts) - 3, txt.count("\n", 0, m.start()) + 1))
    return calls


def _compatible(j, c):
    if j == c:
        return True
    # handles are opaque on the Julia side; a char* handle buffer may be passed as bytes; void* accepts any pointer
    if j.startswith("ptr:") and c.startswith("ptr:") and ("void" in (j[4:], c[4:])):
        return True
    return False


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) >= 60 and "mmm_ctm_fit" in protos and protos["mmm_lda_get"] == ("int", ["ptr:void", "int", "ptr:double", "size_t"])
    calls = shim_ccalls()
    assert len(calls) >= 25
    bad = []
    for name, ret, types, nvals, line in calls:
        if name not in protos:
            bad.append("line %d: %s is not declared in include/mmmusig.h" % (line, name)); continue
        cret, cargs = protos[name]
        if not _compatible(ret, cret):
            bad.append("line %d: %s returns %s, the shim says %s" % (line, name, cret, ret))
        if len(types) != len(cargs):
            bad.append("line %d: %s takes %d arguments, the shim passes %d types" % (line, name, len(cargs), len(types))); continue
        if nvals != len(types):
            bad.append("line %d: %s: %d argument types but %d values" % (line, name, len(types), nvals))
        for i, (j, c) in enumerate(zip(types, cargs)):
            if not _compatible(j, c):
                bad.append("line %d: %s argument %d is %s in the header, %s in the shim" % (line, name, i + 1, c, j))
    assert not bad, "\n".join(bad)


def test_shim_keeps_the_reference_api_surface():
    """exports of MultiModalMuSig.jl:9 plus the fit! keyword arguments of LDA.jl:198, MMCTM.jl:457-458, IMMCTM.jl:437"""
    txt = open(SHIM).read()
    exp = re.search(r"^export\s+(.+)$", txt, flags=re.M)
    assert exp
    names = {n.strip() for n in exp.group(1).split(",")}
    assert {"LDA", "MMCTM", "IMMCTM", "ILDA", "fit!"} <= names
    # the reference's keywords with the reference's defaults, in the reference's order (`resident` is this shim's one addition, default off)
    assert re.search(r"function fit!\(model::LDA; maxiter=1000, tol=1e-4, verbose=true, resident=false\)", txt)
    assert re.search(r"function fit!\(model::ILDA; maxiter=1000, tol=1e-4, verbose=true, resident=false\)", txt)
    assert re.search(r"function fit!\(model::MMCTM; maxiter=100, tol=1e-4, verbose=true, autoα=false, updateΣ=true, resident=false\)", txt)
    assert re.search(r"function fit!\(model::IMMCTM; maxiter=100, tol=1e-4, verbose=true, autoα=false, resident=false\)", txt)


def test_python_binding_table_matches_the_header(mmm):
    """the executable binding (multimodalmusig.jl_amd/_lib.py) declares the same arities as the header"""
    protos = header_prototypes()
    sigs = mmm._lib._SIGS
    assert set(sigs) == set(protos)
    for name, (ret, args) in sigs.items():
        assert len(args) == len(protos[name][1]), "%s: _SIGS has %d arguments, the header %d" % (name, len(args), len(protos[name][1]))


# ---- the API surface the reference's OWN tests drive (tests/golden/reference_test_api.json, made by tests/golden/make_test_api.py from
# /root/reference/test/*.jl -- a name list, not the files) must exist in the shim, method by method ----------------------------------
_IDENT = r"[^\W\d][\w!]*"


def _shim_methods():
    """name -> list of (n_required_positional, n_positional, kw names, has_kw_splat, body)"""
    txt = open(SHIM, encoding="utf-8").read()
    txt = "\n".join(re.sub(r"(?<!\")#(?![^\"]*\"\s*[,)]).*$", "", ln) for ln in txt.split("\n"))      # comments (not '#' inside strings)
    methods = {}
    pat = re.compile(r"(?m)^[ \t]*(?:function[ \t]+)?(" + _IDENT + r")\(")
    for m in pat.finditer(txt):
        name = m.group(1)
        is_fn = txt[m.start():m.end()].lstrip().startswith("function")
        end = _balanced(txt, m.end() - 1)
        rest = txt[end:end + 400]
        if not is_fn and not re.match(r"\s*=(?!=)", rest):
            continue                                      # a call at the start of a line, not a definition
        if is_fn:
            stop = re.search(r"(?m)^" + re.escape(re.match(r"[ \t]*", txt[m.start():]).group(0)) + r"end\b", txt[end:])
            body = txt[end:end + (stop.start() if stop else 0)]
        else:
            nxt = re.search(r"\n(?=\S)", txt[end:])
            body = txt[end:end + (nxt.start() if nxt else len(txt) - end)]
        sig = txt[m.end():end - 1]
        pos, kws, splat, after = [], [], False, False
        depth, cur, parts = 0, "", []
        for ch in sig:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if ch in ",;" and depth == 0:
                parts.append((cur.strip(), ch)); cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append((cur.strip(), ""))
        for arg, sep in parts:
            if after:
                if arg.endswith("..."):
                    splat = True
                else:
                    kws.append(re.match(_IDENT, arg).group(0))
            else:
                pos.append(arg)
            if sep == ";":
                after = True
        nreq = sum(1 for a in pos if not re.search(r"(?<![=!<>])=(?!=)", a))
        methods.setdefault(name, []).append((nreq, len(pos), kws, splat, body))
    return methods


def _reaches_ccall(name, methods, seen=None):
    seen = seen or set()
    if name in seen or name not in methods:
        return False
    seen.add(name)
    for _, _, _, _, body in methods[name]:
        if "ccall((:mmm_" in body:
            return True
        for callee in set(re.findall(r"\b(" + _IDENT + r")\(", body)):
            if callee != name and _reaches_ccall(callee, methods, seen):
                return True
    return False


def test_every_function_the_reference_tests_call_is_defined_in_the_shim():
    import json
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_test_api.json"), encoding="utf-8"))
    assert len(api["calls"]) >= 30
    methods = _shim_methods()
    missing = []
    for name, info in api["calls"].items():
        if name not in methods:
            missing.append("%s (%s)" % (name, info["sites"][0])); continue
        for sig in info["signatures"]:
            ok = any(nreq <= sig["npos"] <= npos and (splat or set(sig["kw"]) <= set(kws)) for nreq, npos, kws, splat, _ in methods[name])
            if not ok:
                missing.append("%s with %d positional arguments and keywords %s (%s)" % (name, sig["npos"], sig["kw"], info["sites"][0]))
        if not _reaches_ccall(name, methods):
            missing.append("%s never reaches a ccall into libmmmusig_hip" % name)
    assert not missing, "the reference's tests call, the shim lacks:\n  " + "\n  ".join(missing)


def test_every_model_field_the_reference_tests_touch_exists_in_the_shim_structs():
    import json
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_test_api.json"), encoding="utf-8"))
    txt = open(SHIM, encoding="utf-8").read()
    for kind, names in api["fields"].items():
        m = re.search(r"mutable struct " + kind + r"\n(.*?)\n\n?    function " + kind, txt, flags=re.S)
        assert m, "no `mutable struct %s` in the shim" % kind
        declared = set(re.findall(r"(" + _IDENT + r")::", m.group(1)))
        lacking = [n for n in names if n not in declared]
        assert not lacking, "%s lacks the fields %s that the reference's tests use" % (kind, lacking)


def test_stage_calls_go_through_upload_and_write_back():
    """every stage function syncs the Julia arrays to the device first (the reference's tests assign fields in place before the call)"""
    methods = _shim_methods()
    for helper in ("lda_stage!", "ctm_stage!", "ctm_doc_stage!", "elbo_terms", "calculate_sumθ", "calculate_Ndivζ"):
        assert helper in methods, helper
        assert all("upload!(model)" in body for _, _, _, _, body in methods[helper]), "%s does not upload the model's arrays before its ccall" % helper
    for name in ("update_ζ!", "update_θ!", "update_ν!", "update_λ!"):      # per-document forms: (model, d), document d only
        assert any(npos == 2 and "ctm_doc_stage!" in body for _, npos, _, _, body in methods[name]), name


# ---- data flow: which fields every entry point uploads BEFORE its ccall and downloads AFTER it -------------------------------------------
# The reference's functions work on the Julia arrays themselves, so a caller may assign any field and call any function
# (scripts/run_mmctm.jl:124-131 assigns γ[m] / Elnϕ[m] / ϕ[m] and calls fit!).  Per reference function: the fields it READS BEFORE IT WRITES
# THEM and the fields it WRITES, read off the reference's source (file:line in the comments).  The shim must upload a superset of the first
# and download a superset of the second.  (Round 4's shim passed every name / arity / prototype check while fit! uploaded nothing.)
_ALL = "ALL"
REF_DATA_FLOW = {
    # LDA.jl / ILDA.jl (same functions on both)
    ("LDA", "update_ϕ!"): ({"Elnθ", "Elnβ"}, {"ϕ"}),                    # LDA.jl:69-76
    ("LDA", "update_Elnθ!"): ({"γ"}, {"Elnθ"}),                          # :78-80
    ("LDA", "update_γ!"): ({"α", "ϕ"}, {"γ", "Elnθ"}),                   # :82-90
    ("LDA", "update_θ!"): ({"γ"}, {"θ"}),                                # :92-94
    ("LDA", "update_Elnβ!"): ({"λ"}, {"Elnβ"}),                          # :96-98
    ("LDA", "update_λ!"): ({"η", "ϕ"}, {"λ", "Elnβ"}),                   # :100-108
    ("LDA", "update_β!"): ({"λ"}, {"β"}),                                # :110-112
    ("LDA", "calculate_elbo"): ({"α", "η", "λ", "Elnβ", "γ", "Elnθ", "ϕ"}, set()),      # :114-172
    # fit! LDA.jl:198-224: update_γ! first (reads ϕ, α), update_ϕ! (reads Elnβ; Elnθ was just written), update_λ! (reads η; ϕ just written)
    ("LDA", "fit!"): ({"α", "η", "Elnβ", "ϕ"}, _ALL),
    ("ILDA", "fit!"): ({"α", "η", "Elnβ", "ϕ"}, _ALL),                   # ILDA.jl:246-272
    # MMCTM.jl / IMMCTM.jl
    ("CTM", "update_ζ!"): ({"λ", "ν"}, {"ζ"}),                           # MMCTM.jl:172-181
    ("CTM", "update_θ!"): ({"λ", "Elnϕ"}, {"θ"}),                        # :183-198
    ("CTM", "update_ν!"): ({"ν", "λ", "ζ", "μ", "invΣ"}, {"ν"}),         # :156-170, common.jl:25-36
    ("CTM", "update_λ!"): ({"λ", "ν", "ζ", "θ", "μ", "invΣ"}, {"λ"}),    # :127-143, common.jl:11-23
    ("CTM", "update_μ!"): ({"λ"}, {"μ"}),                                # :200-202
    ("CTM", "update_Σ!"): ({"λ", "ν", "μ"}, {"Σ", "invΣ"}),              # :204-212
    ("CTM", "update_Elnϕ!"): ({"γ"}, {"Elnϕ"}),                          # :214-222
    ("CTM", "update_γ!"): ({"α", "θ"}, {"γ", "Elnϕ"}),                   # :224-242
    ("CTM", "update_α!"): ({"α", "Elnϕ"}, {"α"}),                        # :252-269
    ("MMCTM", "update_ϕ!"): ({"γ"}, {"ϕ"}),                              # :244-250
    ("MMCTM", "update_props!"): ({"λ"}, {"props"}),                      # :145-154
    ("CTM", "calculate_sumθ"): ({"θ"}, set()),                           # :110-117
    ("CTM", "calculate_Ndivζ"): ({"ζ"}, set()),                          # :119-125
    ("CTM", "calculate_elbo"): ({"α", "μ", "Σ", "invΣ", "γ", "Elnϕ", "λ", "ν", "ζ", "θ"}, set()),      # :271-382
    # fit! MMCTM.jl:457-494: fitdoc! reads λ, ν (ζ, and the start points of both solves), Elnϕ (θ), μ, invΣ (objectives); update_γ! reads α; with
    # updateΣ = false Σ / invΣ are never written and the ELBO reads them; γ / ϕ travel with the Elnϕ a caller seeds (run_mmctm.jl:126-128)
    ("MMCTM", "fit!"): ({"α", "μ", "Σ", "invΣ", "γ", "Elnϕ", "ϕ", "λ", "ν"}, _ALL),
    ("IMMCTM", "fit!"): ({"α", "μ", "Σ", "invΣ", "γ", "Elnϕ", "λ", "ν"}, _ALL),      # IMMCTM.jl:437-466
}
_STRUCT_FIELDS = {"LDA": {"α", "η", "λ", "Elnβ", "β", "γ", "Elnθ", "θ", "ϕ"},
                  "MMCTM": {"α", "μ", "Σ", "invΣ", "γ", "Elnϕ", "ϕ", "λ", "ν", "ζ", "props", "θ"},
                  "IMMCTM": {"α", "μ", "Σ", "invΣ", "γ", "Elnϕ", "λ", "ν", "ζ", "θ"}}


def _sym_tuple(text):
    return set(re.findall(r":(" + _IDENT + r")", text))


def _shim_constants():
    txt = open(SHIM, encoding="utf-8").read()
    out = {}
    for name in ("LDA_ALL_FIELDS", "MMCTM_ALL_FIELDS", "IMMCTM_ALL_FIELDS"):
        m = re.search(r"const " + name + r" = \(([^)]*)\)", txt)
        assert m, name
        out[name] = _sym_tuple(m.group(1))
    m = re.search(r"const FIT_READS = \((.*?)\)\nfunction", txt, flags=re.S)
    assert m, "FIT_READS"
    out["FIT_READS"] = {k: _sym_tuple(v) for k, v in re.findall(r"(\w+) = \(([^)]*)\)", m.group(1))}
    return out


def _typed_methods(name, kind):
    """bodies + signatures of the methods of `name` whose first argument is `model::<one of kinds>`"""
    txt = open(SHIM, encoding="utf-8").read()
    txt = "\n".join(re.sub(r"(?<!\")#(?![^\"]*\"\s*[,)]).*$", "", ln) for ln in txt.split("\n"))
    out = []
    for m in re.finditer(r"(?m)^(function[ \t]+)?" + re.escape(name) + r"\(model::(\w+)", txt):
        if m.group(2) not in kind:
            continue
        end = _balanced(txt, txt.index("(", m.start()))
        if m.group(1):
            stop = re.search(r"(?m)^end\b", txt[end:])
            body = txt[end:end + stop.start()]
        else:
            body = txt[end:txt.index("\n", end)]
        out.append((m.group(2), txt[m.start():end], body))
    return out


def _flow_of(body, kind, consts, methods):
    """(uploaded, downloaded) of one method body; helpers (lda_stage!, ctm_stage!, ctm_doc_stage!, elbo_terms) are followed one level"""
    allf = consts["LDA_ALL_FIELDS"] if kind in ("LDA", "ILDA", "TopicModel") else (
        consts["IMMCTM_ALL_FIELDS"] if kind == "IMMCTM" else consts["MMCTM_ALL_FIELDS"])
    first_ccall = body.find("ccall(")
    up, down = set(), set()
    if "upload_fit_reads!(model)" in body:
        assert 0 <= body.index("upload_fit_reads!(model)") < first_ccall, "upload after the ccall"
        up |= consts["FIT_READS"][kind]
    if "upload!(model)" in body:
        up |= allf
    m = re.search(r"(lda_stage!|ctm_stage!)\(model, .*?, \"[^\"]*\", \(([^)]*)\)\)", body, flags=re.S)
    if m:
        hb = methods[m.group(1)][0][4]
        assert hb.index("upload!(model)") < hb.index("rc_of_call()") < hb.index("download_field!"), m.group(1)
        up |= allf; down |= _sym_tuple(m.group(2))
    m = re.search(r"ctm_doc_stage!\(model, \d, d, \"[^\"]*\", :(" + _IDENT + r")\)", body)
    if m:
        hb = methods["ctm_doc_stage!"][0][4]
        assert hb.index("upload!(model)") < hb.index("ccall(") < hb.index("download_doc!"), "ctm_doc_stage!"
        up |= allf; down.add(m.group(1))
    if "elbo_terms(model)" in body:
        hb = [b for _, _, _, _, b in methods["elbo_terms"]]
        assert all(b.index("upload!(model)") < b.index("ccall(") for b in hb)
        up |= allf
    if "download!(model)" in body:
        assert body.rindex("download!(model)") > first_ccall >= 0, "download before the ccall"
        down |= allf
    return up, down


def test_every_entry_point_uploads_what_the_reference_function_reads_and_downloads_what_it_writes():
    consts = _shim_constants()
    methods = _shim_methods()
    assert consts["LDA_ALL_FIELDS"] == _STRUCT_FIELDS["LDA"] and consts["MMCTM_ALL_FIELDS"] == _STRUCT_FIELDS["MMCTM"] and \
        consts["IMMCTM_ALL_FIELDS"] == _STRUCT_FIELDS["IMMCTM"]
    bad = []
    for (kind, name), (reads, writes) in REF_DATA_FLOW.items():
        kinds = {"LDA": ("LDA", "TopicModel"), "ILDA": ("ILDA", "TopicModel"), "CTM": ("CTM",), "MMCTM": ("MMCTM",), "IMMCTM": ("IMMCTM",)}[kind]
        found = _typed_methods(name, kinds)
        if name == "fit!":
            found = [f for f in found if f[0] == kind]
        if not found:
            bad.append("%s(::%s): no method" % (name, kind)); continue
        for k, sig, body in found:
            up, down = _flow_of(body, kind if kind != "CTM" else "MMCTM", consts, methods)
            if not reads <= up:
                bad.append("%s(::%s) reads %s upstream, the shim uploads only %s before its ccall" % (name, k, sorted(reads), sorted(up)))
            w = _STRUCT_FIELDS["LDA" if kind in ("LDA", "ILDA") else ("IMMCTM" if kind == "IMMCTM" else "MMCTM")] if writes == _ALL else writes
            if kind == "CTM":
                w = w - {"ϕ", "props"}
            if not w <= down:
                bad.append("%s(::%s) writes %s upstream, the shim downloads only %s after its ccall" % (name, k, sorted(w), sorted(down)))
    assert not bad, "\n".join(bad)


def test_fit_uploads_nothing_it_does_not_need_and_the_seeded_restart_goes_through_it():
    """fit! must not push ζ / θ / props (written before read: a K x nnz upload for nothing), and `seed_and_fit_restart` is the reference's
    (scripts/run_mmctm.jl:113-134): item-assign γ[m], Elnϕ[m], ϕ[m], then fit!(tol = 1e-5) WITHOUT `resident`"""
    consts = _shim_constants()
    for kind in ("MMCTM", "IMMCTM"):
        assert not consts["FIT_READS"][kind] & {"ζ", "θ", "props"}
    assert not consts["FIT_READS"]["LDA"] & {"γ", "Elnθ", "θ", "λ", "β"}
    methods = _shim_methods()
    for fn in ("fit_restart", "pick_optimal_modality_models", "fit_seed_models", "seed_and_fit_restart", "pick_optimal_model", "seed_and_fit_model", "fit_model"):
        assert fn in methods, "scripts/run_mmctm.jl's %s has no counterpart in the shim" % fn
    body = methods["seed_and_fit_restart"][0][4]
    for f in ("γ", "Elnϕ", "ϕ"):
        assert re.search(r"model\." + f + r"\[m\] = deepcopy\(opt_models\[m\]\." + f + r"\[m\]\)", body), f
    call = re.search(r"fit!\(model, ([^)]*)\)", body)
    assert call and "tol=1e-5" in call.group(1) and "maxiter=1000" in call.group(1) and "resident" not in call.group(1)
    assert body.index("deepcopy") < body.index("fit!(model")
    # fit_model's positional signature is the script's (run_mmctm.jl:163)
    assert any(npos == 8 for _, npos, _, _, _ in methods["fit_model"])


def test_field_ids_of_upload_and_download_match_the_header():
    """`upload_field!` / `download_field!` address every field by the id the header's enum gives it"""
    hdr = open(HDR).read()
    ids = {}
    m = re.search(r"MMM_CTM_MU = 0.*?MMM_CTM_ALPHA = (\d+)", hdr, flags=re.S)
    for name, val in re.findall(r"MMM_CTM_(\w+) = (\d+)", hdr):
        ids[name] = int(val)
    jl = {"μ": "MU", "Σ": "SIGMA", "invΣ": "INVSIGMA", "γ": "GAMMA", "Elnϕ": "ELNPHI", "ϕ": "PHI", "λ": "LAMBDA", "ν": "NU", "ζ": "ZETA", "props": "PROPS",
          "θ": "THETA", "α": "ALPHA"}
    txt = open(SHIM, encoding="utf-8").read()
    up = txt[txt.index("function upload_field!(model::CTM"):]
    up = up[:up.index("\nend\n")]
    for f, cname in jl.items():
        mm = re.search(r"f == :" + f + r"\n\s*ctm_set\(model, (\d+),", up)
        assert mm and int(mm.group(1)) == ids[cname], "upload_field!(:%s) does not use MMM_CTM_%s = %d" % (f, cname, ids[cname])
    dn = txt[txt.index("function download_field!(model::CTM"):]
    dn = dn[:dn.index("\nend\n")]
    for f, cname in jl.items():
        if f in ("γ", "Elnϕ", "ϕ"):
            continue                      # addressed through `id = f == :γ ? 3 : (f == :Elnϕ ? 4 : 5)`
        mm = re.search(r"f == :" + f + r"\n(.*?)(?=elseif|else\n)", dn, flags=re.S)
        assert mm and re.search(r"ctm_get\(model, %d," % ids[cname], mm.group(1)), "download_field!(:%s)" % f
    assert "id = f == :γ ? %d : (f == :Elnϕ ? %d : %d)" % (ids["GAMMA"], ids["ELNPHI"], ids["PHI"]) in dn
    lda_ids = {n: int(v) for n, v in re.findall(r"MMM_LDA_(\w+) = (\d+)", hdr)}
    mm = re.search(r"const LDA_FIELDS = \(([^)]*)\)", txt)
    got = {k: int(v) for k, v in re.findall(r"(" + _IDENT + r")=(\d+)", mm.group(1))}
    assert got == {"λ": lda_ids["LAMBDA"], "Elnβ": lda_ids["ELNBETA"], "β": lda_ids["BETA"], "γ": lda_ids["GAMMA"], "Elnθ": lda_ids["ELNTHETA"], "θ": lda_ids["THETA"],
                   "ϕ": lda_ids["PHI"]}
