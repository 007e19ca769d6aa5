"""Container-side generator of tests/golden/format_golden.json: the on-disk formats of SURVEY.md 8(f) rank 4, read out
of the REFERENCE's own source text (scene/gaussian_model.py), so that the product's PLY layout and checkpoint tuple are
pinned to the reference rather than to a builder-authored known-answer test.

  * PLY vertex properties: GaussianModel.construct_list_of_attributes (:244-256) is extracted with `ast` and executed
    against a stub `self` carrying the tensor shapes of an SH-degree-3 model; the per-property dtype and the column
    order of the record are read from save_ply (:281-295): the `(attribute, 'f4')` literal and the np.concatenate tuple.
  * checkpoint tuple: the element names of capture()'s returned tuple (:88-104) and of restore()'s unpacking (:106-120).

Only data (names, order, dtype) is written; no reference source text is stored.  Run here (the reference tree does not
exist on the GPU box):  python tests/golden/make_format_golden.py
"""
import ast
import json
import os
import textwrap
import types

REF = "/root/reference/scene/gaussian_model.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "format_golden.json")


def method(tree, name):
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "GaussianModel":
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == name:
                    return f
    raise SystemExit(f"{name} not found in {REF}")


def attr_name(node):
    """self._xyz -> '_xyz'; self._deformation.state_dict() -> '_deformation.state_dict()'; a bare name -> itself."""
    if isinstance(node, ast.Call):
        return attr_name(node.func) + "()"
    if isinstance(node, ast.Attribute):
        base = attr_name(node.value)
        return node.attr if base == "self" else base + "." + node.attr
    if isinstance(node, ast.Name):
        return node.id
    raise ValueError(ast.dump(node))


def main():
    src = open(REF).read()
    tree = ast.parse(src)
    # --- PLY property list: run the reference's own function on the shapes of an SH-degree-3 model
    f = method(tree, "construct_list_of_attributes")
    code = textwrap.dedent(ast.get_source_segment(src, f))
    ns = {}
    exec(compile(code, REF, "exec"), ns)
    shape = lambda *s: types.SimpleNamespace(shape=s)
    stub = types.SimpleNamespace(_features_dc=shape(7, 1, 3), _features_rest=shape(7, 15, 3), _scaling=shape(7, 3), _rotation=shape(7, 4))
    props = ns["construct_list_of_attributes"](stub)
    # --- dtype and column order of a record, from save_ply
    sp = method(tree, "save_ply")
    dtypes, concat = set(), None
    for node in ast.walk(sp):
        if isinstance(node, ast.Tuple) and len(node.elts) == 2 and isinstance(node.elts[1], ast.Constant) and isinstance(node.elts[1].value, str) \
                and isinstance(node.elts[0], ast.Name) and node.elts[0].id == "attribute":
            dtypes.add(node.elts[1].value)
        if isinstance(node, ast.Call) and attr_name(node.func) == "np.concatenate()"[:-2]:
            concat = [attr_name(e) for e in node.args[0].elts]
    assert len(dtypes) == 1 and concat, (dtypes, concat)
    element = None
    for node in ast.walk(sp):
        if isinstance(node, ast.Call) and attr_name(node.func) == "PlyElement.describe":
            element = node.args[1].value
    # --- checkpoint tuple
    cap = method(tree, "capture")
    ret = next(n for n in ast.walk(cap) if isinstance(n, ast.Return))
    capture = [attr_name(e) for e in ret.value.elts]
    res = method(tree, "restore")
    unpack = next(n for n in ast.walk(res) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Tuple))
    restore = [attr_name(e) for e in unpack.targets[0].elts]
    out = {"source": "scene/gaussian_model.py of the reference: construct_list_of_attributes, save_ply, capture, restore",
           "ply": {"element": element, "properties": props, "dtype": dtypes.pop(), "record_columns": concat,
                   "sh_degree": 3, "n_properties": len(props)},
           "checkpoint": {"capture": capture, "restore_targets": restore, "length": len(capture)}}
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
