"""The Rust binding (bindings/rust/) cannot be compiled here (no cargo / rustc in the image): check mechanically that
its `extern "C"` block declares exactly the symbols of include/tinyrenderer_hip.h with the header's number of
arguments, that the facade only calls symbols that exist, and that it has no stub bodies."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _header_functions():
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "tinyrenderer_hip.h")).read())
    out = {}
    for name, args in re.findall(r"\b(tr_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_functions():
    text = open(os.path.join(ROOT, "bindings", "rust", "tinyrenderer-hip-sys", "src", "lib.rs")).read()
    text = re.sub(r"//.*", "", text)
    block = text[text.index('extern "C" {'):]
    out = {}
    for name, args in re.findall(r"pub fn (tr_[a-z0-9_]+)\s*\(([^)]*)\)", block, flags=re.S):
        args = args.strip()
        out[name] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_extern_block_matches_the_header():
    header, rust = _header_functions(), _rust_functions()
    assert len(header) >= 45
    assert sorted(rust) == sorted(header), (sorted(set(header) - set(rust)), sorted(set(rust) - set(header)))
    assert {n: rust[n] for n in header} == header


def test_draw_params_layout_matches_the_header():
    """Field order of tr_draw_params / tr_shade_ext in the -sys crate = the header's (both are #[repr(C)])."""
    header = _strip_c_comments(open(os.path.join(ROOT, "include", "tinyrenderer_hip.h")).read())
    rust = re.sub(r"//.*", "", open(os.path.join(ROOT, "bindings", "rust", "tinyrenderer-hip-sys", "src", "lib.rs")).read())
    for name in ("tr_draw_params", "tr_shade_ext"):
        c_body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, flags=re.S).group(1)
        c_fields = [re.sub(r"\[.*", "", f.strip().split()[-1]).lstrip("*") for f in c_body.split(";") if f.strip()]
        r_body = re.search(r"pub struct %s \{(.*?)\n\}" % name, rust, flags=re.S).group(1)
        r_fields = re.findall(r"pub ([a-z_0-9]+):", r_body)
        assert r_fields == c_fields, name


def test_facade_is_complete_and_calls_existing_symbols():
    text = open(os.path.join(ROOT, "bindings", "rust", "tinyrenderer-hip", "src", "lib.rs")).read()
    code = re.sub(r"//.*", "", text)
    for word in ("todo!", "unimplemented!", "/* ...", "likewise */"):
        assert word not in code
    used = set(re.findall(r"sys::(tr_[a-z0-9_]+)\s*\(", code))
    assert used and used <= set(_header_functions()), used - set(_header_functions())
    for item in ("pub struct ModelRendererState", "pub fn new(frame_width: usize, frame_height: usize) -> ModelRendererState",
                 "pub fn clear_screen(model_renderer_state: &mut ModelRendererState)", "pub fn render_mesh_component(",
                 "pub fn frame_buffer(", "pub fn z_buffer(", "impl Drop for ModelRendererState"):
        assert item in code, item
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")")
