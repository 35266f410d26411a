"""Writes tests/golden/eagle_tree.json: the inputs and expected outputs that the reference's own self-test of the
draft-tree builder holds (python/sglang/srt/speculative/build_eagle_tree.py:115-384), read from the file's literals
(nothing of the reference is imported or run).  Test infrastructure; run once here, the JSON is committed."""
import ast
import json
import os
import sys

SRC = "/root/reference/python/sglang/srt/speculative/build_eagle_tree.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eagle_tree.json")


class _Torch:
    float32, int32, int64 = "float32", "int32", "int64"

    @staticmethod
    def tensor(data, dtype=None, device=None):
        return {"data": data, "dtype": dtype or "int64"}


def main():
    tree = ast.parse(open(SRC).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "test_build_tree_kernel_efficient")
    ns = {"torch": _Torch}
    wanted = {"verified_id", "score_list", "token_list", "parents_list", "seq_lens", "topk", "depth", "num_draft_token"}
    expected = {}
    for node in fn.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in wanted:
            exec(compile(ast.Module([node], []), SRC, "exec"), ns)
        if isinstance(node, ast.Assert) and isinstance(node.test, ast.Compare):
            left = node.test.left                                   # <name>.tolist() == <literal>
            expected[left.func.value.id] = ast.literal_eval(node.test.comparators[0])
    doc = {k: ns[k] for k in wanted}
    doc["expected"] = expected
    with open(OUT, "w") as f:
        json.dump(doc, f)
    print(OUT, sorted(expected))


if __name__ == "__main__":
    sys.exit(main())
