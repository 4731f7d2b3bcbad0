"""
Static guard on the Julia shim (Julia cannot run in this image): every `ccall((:tmvb_xxx, LIBTMVB), ret, (types...), ...)`
in topicmodelsvb.jl_amd/julia/*.jl must name a function that include/tmvb.h declares, with the same number of arguments
and compatible argument classes (integer width, double, pointer and pointee type), and the status return type.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_CLASS = {"int32_t": "i32", "int": "i32", "int64_t": "i64", "double": "f64", "float": "f32", "char": "i8"}
JL_SCALAR = {"Int32": "i32", "Cint": "i32", "Int64": "i64", "Float64": "f64", "Cdouble": "f64", "Cfloat": "f32", "Float32": "f32",
             "Cchar": "i8", "UInt8": "u8"}


def header_protos():
    hdr = open(os.path.join(ROOT, "include", "tmvb.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"typedef int \(\*tmvb_host_allreduce_fn\)\([^;]*;", "", hdr)
    protos = {}
    for ret, name, args in re.findall(r"\b(int|void|const char\*)\s+(tmvb_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        params = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\[[^\]]*\]", "*", a)
                stars = a.count("*")
                base = re.sub(r"\bconst\b", "", a.replace("*", " ")).split()
                # drop the parameter name
                ty = base[0] if len(base) >= 1 else ""
                if base[0] in ("unsigned", "struct"):
                    ty = base[1]
                if stars == 0:
                    if ty == "tmvb_host_allreduce_fn":
                        params.append(("ptr", "fn"))
                    else:
                        params.append((C_CLASS[ty], None))
                else:
                    pointee = C_CLASS.get(ty, "u8" if ty == "uint8_t" else "opaque")
                    if stars >= 2:
                        pointee = "ptr"
                    if ty == "void":
                        pointee = "any"
                    params.append(("ptr", pointee))
        protos[name] = (ret, params)
    return protos


def jl_class(t):
    t = t.strip()
    if t in JL_SCALAR:
        return (JL_SCALAR[t], None)
    if t == "Cstring":
        return ("ptr", "i8")
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.*)\}", t)
    assert m, f"unrecognised ccall argument type {t!r}"
    inner = m.group(1).strip()
    if inner == "Cvoid":
        return ("ptr", "any")
    if inner.startswith("Ptr{"):
        return ("ptr", "ptr")
    return ("ptr", JL_SCALAR[inner])


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        if ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def ccalls(src):
    for m in re.finditer(r"ccall\(\(:(tmvb_[a-z0-9_]+),\s*LIBTMVB\),\s*([A-Za-z]+),\s*\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        types = split_top(src[i:j - 1])
        # the call arguments follow up to the matching close of `ccall(`
        k, depth = j, 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[k], 0)
            k += 1
        args = split_top(src[j:k - 1].lstrip(","))
        yield m.group(1), m.group(2), types, args


def compatible(c, j):
    if c[0] != j[0]:
        return False
    if c[0] != "ptr":
        return True
    cp, jp = c[1], j[1]
    if cp in ("any", "fn") or jp == "any":
        return True                      # void* / opaque handles / C_NULL-able
    if cp == "opaque":
        return jp in ("any",)            # struct handles travel as Ptr{Cvoid}
    if cp == "ptr":
        return jp in ("ptr", "any")
    if cp == "u8" or cp == "i8":
        return jp in ("u8", "i8")
    return cp == jp


def test_every_ccall_matches_the_header():
    protos = header_protos()
    jdir = os.path.join(ROOT, "topicmodelsvb.jl_amd", "julia")
    seen = set()
    n = 0
    for f in sorted(os.listdir(jdir)):
        if not f.endswith(".jl"):
            continue
        src = open(os.path.join(jdir, f)).read()
        for name, ret, types, args in ccalls(src):
            n += 1
            assert name in protos, f"{f}: ccall of {name}, which include/tmvb.h does not declare"
            cret, params = protos[name]
            if cret == "int":
                assert ret == "Cint", f"{f}: {name} returns a status (int), ccall says {ret}"
            elif cret == "const char*":
                assert ret == "Cstring", (name, ret)
            assert len(types) == len(params), f"{f}: {name} takes {len(params)} arguments, the ccall type tuple has {len(types)}: {types}"
            assert len(args) == len(params), f"{f}: {name} takes {len(params)} arguments, the ccall passes {len(args)}: {args}"
            for q, (c, t) in enumerate(zip(params, types)):
                assert compatible(c, jl_class(t)), f"{f}: {name} argument {q}: header {c}, ccall {t}"
            seen.add(name)
    assert n >= 40
    # the operators a maintainer needs are all bound
    for must in ("tmvb_lda_train", "tmvb_ctm_train", "tmvb_ctpf_train", "tmvb_lda_train_group", "tmvb_ctm_train_group", "tmvb_comm_create_rccl",
                 "tmvb_comm_create_rccl_all", "tmvb_comm_create_host", "tmvb_comm_unique_id", "tmvb_lda_set_comm", "tmvb_ctm_set_comm",
                 "tmvb_ctpf_set_comm", "tmvb_ctpf_set_state_old", "tmvb_ctpf_recommend", "tmvb_ctm_update_sigma", "tmvb_ctm_update_mu",
                 "tmvb_ctpf_mstep", "tmvb_lda_estep", "tmvb_ctm_estep", "tmvb_ctpf_estep", "tmvb_flda_train", "tmvb_fctm_train",
                 "tmvb_flda_set_state", "tmvb_fctm_set_state", "tmvb_flda_get_state", "tmvb_fctm_get_state", "tmvb_flda_estep", "tmvb_fctm_estep",
                 "tmvb_flda_update_eta", "tmvb_fctm_update_sigma", "tmvb_flda_train_group", "tmvb_fctm_train_group", "tmvb_flda_set_comm",
                 "tmvb_fctm_set_comm"):
        assert must in seen, f"the Julia shim does not bind {must}"


def test_header_parser_sees_the_whole_abi(tmvb):
    protos = header_protos()
    assert set(protos) == set(tmvb.exported_symbols())
    assert protos["tmvb_lda_train"][1][-1] == ("ptr", "f64") and len(protos["tmvb_lda_train"][1]) == 11
    assert protos["tmvb_corpus_create"][1][1] == ("i64", None)


def test_gpu_macro_file_covers_the_models():
    src = open(os.path.join(ROOT, "topicmodelsvb.jl_amd", "julia", "gpu_macro.jl")).read()
    for t in ("LDA", "CTM", "CTPF", "fLDA", "fCTM"):
        assert f"copyback!(model::{t}, dev::hip{t})" in src and f"hipmodel(model::{t}) = hip{t}(model)" in src
    assert "macro gpu(expr::Expr)" in src


def _ref_assignments(lo, hi):
    """`model.<field> = ...` statements of the reference's @gpu copy-back (src/macros.jl lines lo..hi), as written there.
    The list is data about the reference's interface, spelled out here because /root/reference does not travel."""
    return {
        (136, 149): ["topics", "alpha", "beta", "Elogtheta", "Elogtheta_old", "gamma", "phi", "elbo", "beta_old"],
        (177, 192): ["topics", "mu", "sigma", "invsigma", "beta", "lambda", "lambda_old", "vsq", "logzeta", "phi", "elbo", "beta_old"],
        (239, 265): ["topics", "scores", "drecs", "urecs", "alef", "alef_old", "he", "he_old", "bet", "bet_old", "vav", "vav_old", "gimel",
                     "gimel_old", "zayin", "zayin_old", "dalet", "dalet_old", "het", "het_old", "phi", "xi", "elbo"],
    }[(lo, hi)]


def _function_body(src, signature):
    i = src.index(signature)
    j = src.index("\nend\n", i)
    return src[i:j]


def test_copyback_assigns_every_field_the_reference_assigns_from_the_device():
    """Every field the reference's `@gpu` assigns back to the host model (src/macros.jl:136-149 LDA, :177-192 CTM, :239-265
    CTPF) is assigned in the matching copyback!; state fields come straight from `dev.`, `*_old` fields are copies of the
    freshly assigned host field (as in the reference), and phi / xi are rebuilt by dev_phi1 / dev_xi1, whose bodies read the
    device's *_old state only (the shim once rebuilt phi from the HOST model's pre-training beta_old)."""
    src = open(os.path.join(ROOT, "topicmodelsvb.jl_amd", "julia", "gpu_macro.jl")).read()
    for T, span in (("LDA", (136, 149)), ("CTM", (177, 192)), ("CTPF", (239, 265))):
        body = _function_body(src, f"function copyback!(model::{T}, dev::hip{T})")
        assigns = dict(re.findall(r"^\s*model\.(\w+)\s*=\s*([^#\n]+)", body, flags=re.M))
        for field in _ref_assignments(*span):
            assert field in assigns, f"copyback!(::{T}) does not assign model.{field} (src/macros.jl:{span[0]}-{span[1]})"
            rhs = assigns[field].strip()
            if field in ("phi", "xi"):
                assert rhs == f"dev_{field}1(dev)", (T, field, rhs)
            elif field.endswith("_old"):
                base = field[:-4]
                assert re.fullmatch(rf"(copy|deepcopy)\(model\.{base}\)", rhs), (T, field, rhs)
                # ... and the copy is taken after the field itself was assigned from the device
                assert body.index(f"model.{base} =") < body.index(f"model.{field} ="), (T, field)
            else:
                assert re.fullmatch(rf"(Symmetric\()?dev\.{field}\)?", rhs), (T, field, rhs)
    # phi / xi for CTPF are taken from dev.*_old BEFORE any host *_old is overwritten, and the builders never touch `model`
    for fn in ("dev_phi1(dev::hipLDA)", "dev_phi1(dev::hipCTM)", "dev_phi1(dev::hipCTPF)", "dev_xi1(dev::hipCTPF)", "dev_phi1(dev::hipfLDA)",
               "dev_phi1(dev::hipfCTM)"):
        body = _function_body(src, "function " + fn)
        assert "model." not in body, fn
        assert "_old" in body, fn
    assert "host_phi1" not in src
