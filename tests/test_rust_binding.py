"""The Rust crate (rust/particles-hip) cannot be compiled here (no Rust
toolchain), so its ``extern "C"`` block is checked as text against the header
it binds: every function exists in include/particles_hip.h with the same
number of parameters, compatible parameter and return types, and every struct
it passes by pointer has the header's size.  build.rs must list the sources the
Makefile builds the library from."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def split_params(s):
    """Top-level comma split (parentheses and brackets may nest)."""
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


def c_declarations():
    text = strip_c_comments(open(os.path.join(ROOT, "include", "particles_hip.h")).read())
    decls = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(pcl_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        plist = [] if params in ("", "void") else split_params(params)
        decls[name] = (ret, plist)
    return decls


def c_kind(param):
    """Coarse type class of a C parameter or return type."""
    p = param.strip()
    if "*" in p or "[" in p:
        return "ptr"
    if re.search(r"\b(size_t)\b", p):
        return "usize"
    if re.search(r"\buint64_t\b", p):
        return "u64"
    if re.search(r"\buint32_t\b", p):
        return "u32"
    if re.search(r"\bfloat\b", p):
        return "f32"
    if re.search(r"\bint\b", p):
        return "int"
    if p == "void":
        return "void"
    raise AssertionError(f"unclassified C type: {param!r}")


def rust_kind(t):
    t = t.strip()
    if t.startswith("*"):
        return "ptr"
    return {"usize": "usize", "u64": "u64", "u32": "u32", "f32": "f32", "c_int": "int", "i32": "int", "": "void"}[t]


def rust_declarations():
    text = open(os.path.join(ROOT, "rust", "particles-hip", "src", "lib.rs")).read()
    text = re.sub(r"//.*", "", text)
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    decls = {}
    for m in re.finditer(r"fn\s+(pcl_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, params, ret = m.group(1), " ".join(m.group(2).split()), (m.group(3) or "").strip()
        plist = [p.split(":", 1)[1].strip() for p in split_params(params)] if params else []
        decls[name] = (ret, plist)
    return decls, text


def test_every_rust_extern_matches_the_header():
    c, (r, _) = c_declarations(), rust_declarations()
    assert len(r) >= 40
    for name, (ret, params) in r.items():
        assert name in c, f"{name} is declared in lib.rs but not in particles_hip.h"
        c_ret, c_params = c[name]
        assert len(params) == len(c_params), f"{name}: {len(params)} parameters in Rust, {len(c_params)} in C"
        for i, (rp, cp) in enumerate(zip(params, c_params)):
            assert rust_kind(rp) == c_kind(cp), f"{name}, parameter {i}: Rust {rp!r} vs C {cp!r}"
        want = c_kind(c_ret)
        # `const char*` and `pcl_compute*` returns are pointers
        assert rust_kind(ret) == want, f"{name}: returns {ret!r} in Rust, {c_ret!r} in C"


def test_the_reference_surface_and_the_new_pieces_are_bound():
    r, text = rust_declarations()
    for must in ("pcl_create", "pcl_upload_particles", "pcl_update_force_grid", "pcl_update_sim_params", "pcl_step",
                 "pcl_read_particles", "pcl_diagnostics_read", "pcl_patch_force_grid_indexed", "pcl_group_create",
                 "pcl_group_step", "pcl_group_diagnostics_read", "pcl_poly7_from_points"):
        assert must in r
    # the reference's method names (src/compute.rs:25,182,198,206,221) and the additions
    for method in ("pub fn new<", "pub fn new_on_device<", "pub fn new_sharded<", "pub fn upload_particles<",
                   "pub fn update_force_grid(", "pub fn update_sim_params<", "pub fn compute(", "pub fn diagnostics(",
                   "pub fn patch_force_grid_indexed("):
        assert method in text, method
    assert "device as c_int" in text          # the device ordinal reaches pcl_create


def test_rust_struct_layouts_match_the_header():
    _, text = rust_declarations()
    sizes = {"f32": 4, "i32": 4, "u32": 4, "f64": 8}

    def rust_size(name):
        body = re.search(r"pub struct " + name + r"\s*\{(.*?)\n\}", text, flags=re.S).group(1)
        total = 0
        for m in re.finditer(r"pub\s+\w+\s*:\s*(\[(\w+);\s*(\d+)\]|\w+)", body):
            if m.group(2):
                total += sizes[m.group(2)] * int(m.group(3))
            else:
                total += sizes[m.group(1)]
        return total

    assert rust_size("BrushInput") == 64          # pcl_brush_input
    assert rust_size("Diagnostics") == 48         # pcl_diagnostics
    assert rust_size("Poly7") == 32               # pcl_poly7


def test_build_script_lists_the_makefile_sources():
    build = open(os.path.join(ROOT, "rust", "particles-hip", "build.rs")).read()
    make = open(os.path.join(ROOT, "Makefile")).read()
    sources = set(re.findall(r"particles_amd/csrc/[a-z_]+\.(?:hip|cpp)", make))
    assert sources and sources == set(re.findall(r'"(particles_amd/csrc/[a-z_]+\.(?:hip|cpp))"', build))
