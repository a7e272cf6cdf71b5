"""Extract, from the reference's Julia sources, the names the Julia binding (julia/ACMEHip.jl) relies on:

  * the variables each element's `nonlinear_eq` closure CAPTURES (src/elements.jl) -- Julia exposes them as
    the closure object's fields, which is how `describe_element` recovers the element parameters --, the
    number of q entries the closure takes and of residuals it returns;
  * the captures of the closures a `DiscreteModel` holds (src/circuit.jl:68-86, src/ACME.jl:176-194,236-252);
  * the field names of the structs the binding reads (`DiscreteModel`, `ParametricNonLinEq`, `CircuitNLFunc`,
    the solver types).

Runs in the BUILD container only (it reads /root/reference, which does not exist on the GPU box) and writes
tests/golden/closure_captures.json -- names and counts, no source text.  tests/test_julia_binding.py checks
the binding against that fixture, and re-runs this extraction when the reference is present.

    python tests/golden/make_closure_captures.py            # regenerate the fixture
"""
import json
import os
import re
import sys

REF = os.environ.get("ACME_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
IDENT = re.compile(r"[^\W\d]\w*", re.UNICODE)
OPENERS = ("function", "if", "for", "while", "let", "begin", "do", "try", "struct", "quote", "macro", "module")


def strip_comments_and_strings(text):
    out = []
    for line in text.split("\n"):
        line = re.sub(r'"(?:[^"\\]|\\.)*"', '""', line)
        out.append(line.split("#", 1)[0])
    return "\n".join(out)


def block_end(text, start):
    """index just past the `end` closing the block keyword that starts at `start`"""
    depth, brackets = 0, 0
    for m in re.finditer(r"[\[\]]|[^\W\d]\w*", text[start:], re.UNICODE):
        tok = m.group(0)
        if tok == "[":
            brackets += 1
        elif tok == "]":
            brackets -= 1
        elif brackets == 0 and tok in OPENERS:
            depth += 1
        elif brackets == 0 and tok == "end":
            depth -= 1
            if depth == 0:
                return start + m.end()
    raise ValueError("unterminated block")


def balanced(text, start, open_="(", close=")"):
    depth = 0
    for i in range(start, len(text)):
        if text[i] == open_:
            depth += 1
        elif text[i] == close:
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def split_top(s, seps=",;"):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch in seps and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    return [p.strip() for p in parts if p.strip()]


def param_names(sig):
    names = []
    for p in split_top(sig):
        m = IDENT.match(p.lstrip(":"))
        if m and not p.startswith("::"):
            names.append(m.group(0))
    return names


def assigned_names(text):
    """names bound by plain / updating assignments, tuple destructuring, `local`, `let` and `for`"""
    names = set()
    for line in text.split("\n"):
        m = re.match(r"^\s*((?:[^\W\d]\w*\s*,\s*)+[^\W\d]\w*)\s*=[^=]", line, re.UNICODE)       # a, b = q
        if m:
            names.update(IDENT.findall(m.group(1)))
        for m in re.finditer(r"(?:^|[\s;(])([^\W\d][\w´₁₂⁻¹]*)\s*(?:[-+*/]?=)(?!=)", line, re.UNICODE):
            names.add(m.group(1))
        m = re.match(r"^\s*local\s+(.*)$", line)
        if m:
            names.update(IDENT.findall(m.group(1)))
    return names


def closure_info(text, enclosing_bound, closure_start, arrow=False):
    """captured names (in order of first use), parameter names, body text of the closure at closure_start"""
    if arrow:                                    # (a, b) -> expr  : body up to the end of the line group
        pend = balanced(text, closure_start)
        params = param_names(text[closure_start + 1:pend - 1])
        body = text[pend:text.index("\n", pend)]
    else:
        p0 = text.index("(", closure_start)
        pend = balanced(text, p0)
        params = param_names(text[p0 + 1:pend - 1])
        body = text[pend:block_end(text, closure_start)]
    local = set(params) | assigned_names(body)
    seen = []
    for m in IDENT.finditer(body):
        n = m.group(0)
        # (a name followed by ´ / subscripts is a different identifier: res´, Jq´, q₁⁻¹ ...)
        nxt = body[m.end():m.end() + 1]
        if nxt in "´⁻":
            continue
        if n in enclosing_bound and n not in local and n not in seen:
            seen.append(n)
    return dict(captures=seen, params=params), body


def element(text, name, sig_regex, extra_bound=()):
    m = re.search(sig_regex, text)
    assert m, name
    sig_end = balanced(text, m.end() - 1)
    bound = set(param_names(text[m.end():sig_end - 1])) | set(extra_bound)
    c0 = re.search(r"function\s*\(q\)", text[sig_end:])
    assert c0, name
    start = sig_end + c0.start()
    bound |= assigned_names(text[sig_end:start])
    info, body = closure_info(text, bound, start)
    # q entries: destructuring `a, b, c = q` or the largest q[i]; residuals: entries of res = @SVector [...]
    m = re.search(r"^\s*((?:[^\W\d]\w*\s*,\s*)*[^\W\d]\w*)\s*=\s*q\s*$", body, re.M | re.UNICODE)
    nq = len(IDENT.findall(m.group(1))) if m else max(int(i) for i in re.findall(r"\bq\[(\d+)\]", body))
    r = re.search(r"res\s*=\s*@SVector\s*\[", body)
    rend = balanced(body, r.end() - 1, "[", "]")
    nn = len(split_top(body[r.end():rend - 1], ","))
    info.update(nq=nq, nn=nn, line=text[:start].count("\n") + 1)
    return info


def struct_fields(text, name):
    m = re.search(r"(?:mutable\s+)?struct\s+" + name + r"\b[^\n]*\n", text)
    assert m, name
    end = block_end(text, m.start() + m.group(0).index("struct"))
    fields = []
    depth = 0
    for line in text[m.end():end].split("\n"):
        s = line.strip()
        toks = IDENT.findall(s)
        if depth == 0 and toks and toks[0] not in OPENERS + ("end", "new", "return"):
            fm = re.match(r"^([^\W\d]\w*)\s*(::|$)", s, re.UNICODE)
            if fm:
                fields.append(fm.group(1))
        for t in toks:
            if t in OPENERS:
                depth += 1
            elif t == "end":
                depth -= 1
    return fields


def extract():
    rd = lambda f: strip_comments_and_strings(open(os.path.join(REF, "src", f), encoding="utf-8").read())  # noqa: E731
    el, ci, ac, so = rd("elements.jl"), rd("circuit.jl"), rd("ACME.jl"), rd("solvers.jl")
    out = {"source": "HSU-ANT/ACME.jl src/{elements,circuit,ACME,solvers}.jl (names and counts only)"}
    out["elements"] = {
        "potentiometer": element(el, "potentiometer", r"\npotentiometer\(r\)\s*=\s*\n?\s*Element\(".replace(r"Element\(", "") and r"\npotentiometer\("),
        "diode": element(el, "diode", r"\n_diode\("),
        "bjt": element(el, "bjt", r"\nfunction __bjt\("),
        "mosfet": element(el, "mosfet", r"\nfunction __mosfet\("),
        "opamp_macak": element(el, "opamp_macak", r"function opamp\(::Type\{Val\{:macak\}\}"[:-0] + r"(?=,)|function opamp\("),
        "transformer_ja": element(el, "transformer_ja", r"\nfunction __transformer_ja\("),
    }
    # the potentiometer with a free position is the SECOND method named potentiometer( (the one-argument form)
    pots = [m for m in re.finditer(r"\npotentiometer\(", el)]
    assert len(pots) == 2
    sig_end = balanced(el, pots[1].end() - 1)
    bound = set(param_names(el[pots[1].end():sig_end - 1]))
    start = sig_end + re.search(r"function\s*\(q\)", el[sig_end:]).start()
    info, body = closure_info(el, bound, start)
    m = re.search(r"^\s*((?:[^\W\d]\w*\s*,\s*)*[^\W\d]\w*)\s*=\s*q\s*$", body, re.M | re.UNICODE)
    r = re.search(r"res\s*=\s*@SVector\s*\[", body)
    rend = balanced(body, r.end() - 1, "[", "]")
    info.update(nq=len(IDENT.findall(m.group(1))), nn=len(split_top(body[r.end():rend - 1], ",")),
                line=el[:start].count("\n") + 1)
    out["elements"]["potentiometer"] = info
    # opamp(Val{:macak}): signature has an unnamed first parameter
    m = re.search(r"function opamp\(::Type\{Val\{:macak\}\}", el)
    sig_end = balanced(el, m.start() + len("function opamp"))
    bound = set(param_names(el[m.start() + len("function opamp("):sig_end - 1]))
    start = sig_end + re.search(r"function\s*\(q\)", el[sig_end:]).start()
    bound |= assigned_names(el[sig_end:start])
    info, body = closure_info(el, bound, start)
    mm = re.search(r"^\s*((?:[^\W\d]\w*\s*,\s*)*[^\W\d]\w*)\s*=\s*q\s*$", body, re.M | re.UNICODE)
    r = re.search(r"res\s*=\s*@SVector\s*\[", body)
    rend = balanced(body, r.end() - 1, "[", "]")
    info.update(nq=len(IDENT.findall(mm.group(1))), nn=len(split_top(body[r.end():rend - 1], ",")),
                line=el[:start].count("\n") + 1)
    out["elements"]["opamp_macak"] = info

    closures = {}
    # src/circuit.jl: let q_indices=..., nleqfunc=...; function (q) nleqfunc(q[q_indices]) end
    f0 = ci.index("function nonlinear_eq_func(c::Circuit")
    l0 = ci.index("let q_indices", f0)
    letline = ci[l0:ci.index("@inline", l0)]
    bound = set(re.findall(r"([^\W\d]\w*)\s*=", letline, re.UNICODE))
    start = l0 + re.search(r"function\s*\(q\)", ci[l0:]).start()
    closures["circuit_nl_item"], _ = closure_info(ci, bound, start)
    closures["circuit_nl_item"]["line"] = ci[:start].count("\n") + 1
    # src/ACME.jl: let q = zeros(nq), circ_nl_func = ...; function(res, J, pfull, Jq, fq, z)
    l0 = ac.index("model_nonlinear_eq_funcs = Function[")
    let0 = ac.index("let ", l0)
    letline = ac[let0:ac.index("@inline", let0)]
    bound = set(re.findall(r"([^\W\d]\w*)\s*=", letline, re.UNICODE))
    start = let0 + re.search(r"function\s*\(res", ac[let0:]).start()
    closures["model_nl_func"], _ = closure_info(ac, bound, start)
    closures["model_nl_func"]["line"] = ac[:start].count("\n") + 1
    # nonlinear_eq_funcs = Function[ function (res, J, scratch, z) nleq(...) end for (nleq, fq) in zip(...)]
    l0 = ac.index("\n    nonlinear_eq_funcs = Function[")
    start = l0 + re.search(r"function\s*\(res", ac[l0:]).start()
    endc = block_end(ac, start)
    forvars = re.search(r"for\s*\(([^)]*)\)\s*in", ac[endc:endc + 200])
    closures["model_nl_wrapper"], _ = closure_info(ac, set(param_names(forvars.group(1))), start)
    closures["model_nl_wrapper"]["line"] = ac[:start].count("\n") + 1
    for key, anchor in (("set_p", "nonlinear_eq_set_ps = ["), ("calc_Jp", "nonlinear_eq_calc_Jps = [")):
        l0 = ac.index(anchor)
        start = l0 + re.search(r"function\s*\(scratch", ac[l0:]).start()
        endc = block_end(ac, start)
        forvars = re.search(r"for\s*(\([^)]*\)|[^\W\d]\w*)\s*in", ac[endc:endc + 200], re.UNICODE)
        closures[key], _ = closure_info(ac, set(IDENT.findall(forvars.group(1))), start)
        closures[key]["line"] = ac[:start].count("\n") + 1
    out["closures"] = closures
    out["structs"] = {
        "DiscreteModel": struct_fields(ac, "DiscreteModel"),
        "CircuitNLFunc": struct_fields(ci, "CircuitNLFunc"),
        "ParametricNonLinEq": struct_fields(so, "ParametricNonLinEq"),
        "SimpleSolver": struct_fields(so, "SimpleSolver"),
        "HomotopySolver": struct_fields(so, "HomotopySolver"),
        "CachingSolver": struct_fields(so, "CachingSolver"),
    }
    # size accessors and generic functions the binding calls / extends
    funcs = {}
    for name in ("nx", "nu", "ny", "nn", "nq", "np"):
        funcs[name] = bool(re.search(r"\n" + name + r"\(model::DiscreteModel", ac)) or bool(re.search(r"\b" + name + r"\(model::DiscreteModel", ac))
    for name in ("solve", "hasconverged", "needediterations", "set_resabstol!", "get_extrapolation_origin",
                 "set_extrapolation_origin", "get_extrapolation_jacobian"):
        funcs[name] = bool(re.search(r"\n(?:function\s+)?" + re.escape(name) + r"\(solver::SimpleSolver", so))
    funcs["run!"] = bool(re.search(r"\nfunction run!\(runner::ModelRunner|\nrun!\(", ac))
    out["functions"] = funcs
    out["default_solver"] = re.search(r"DiscreteModel\(circ::Circuit, t::Real,\s*::Type\{Solver\}=([^;)]*)", ac).group(1).strip()
    return out


if __name__ == "__main__":
    data = extract()
    path = os.path.join(HERE, "closure_captures.json")
    with open(path, "w", encoding="utf-8") as fh:
        json.dump(data, fh, indent=1, ensure_ascii=False, sort_keys=True)
        fh.write("\n")
    json.dump(data, sys.stdout, indent=1, ensure_ascii=False)
