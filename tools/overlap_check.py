"""Share of a file's statements that occur verbatim in the reference's core/*.py (developer / review tool).
Both sides go through `ast` (docstrings and comments dropped, one statement per line), as the round-2 review did.
   python tools/overlap_check.py ics-wt-physicsengine_amd/core/physics.py ...        (needs /root/reference; not run on the GPU box)"""
import ast, glob, os, sys

REF = "/root/reference/src/wt_simulator/core"


def normalised_lines(path):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef, ast.Module)):
            body = node.body
            if body and isinstance(body[0], ast.Expr) and isinstance(getattr(body[0], "value", None), ast.Constant) \
                    and isinstance(body[0].value.value, str):
                node.body = body[1:] or [ast.Pass()]
    return [l.strip() for l in ast.unparse(tree).split("\n") if l.strip()]


if __name__ == "__main__":
    ref = set()
    for f in glob.glob(os.path.join(REF, "*.py")):
        ref.update(normalised_lines(f))
    for path in sys.argv[1:]:
        lines = normalised_lines(path)
        hit = [l for l in lines if l in ref]
        print(f"{path}: {len(lines)} statements, {len(hit)} verbatim in the reference ({100.0 * len(hit) / max(1, len(lines)):.0f} %)")
        if os.environ.get("SHOW"):
            for l in hit:
                print("    ", l)
