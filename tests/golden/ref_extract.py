"""Run statement ranges of the reference's un-importable driver files AS THEY ARE, read from /root/reference at generation time.

``evaluate.py`` cannot be imported (tensorboard, process group, DETR download), but the golden generators need two inline
blocks of ``eval_sgd`` - the object front-end and the pair loop over the predicted objects.  Instead of carrying a restatement of
those lines, the generators locate the block by its first and last line (short anchor strings), dedent it and ``exec`` it in a
namespace that provides the names the block reads.  No reference source is committed; only anchors are."""
import textwrap


def reference_block(path, func, first_line_contains, last_line_contains):
    """Source text of the statements from the first line containing ``first_line_contains`` to the first later line containing
    ``last_line_contains`` (inclusive), inside ``def <func>``; dedented."""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("def " + func + "("))
    a = next(i for i in range(start, len(lines)) if first_line_contains in lines[i])
    b = next(i for i in range(a, len(lines)) if last_line_contains in lines[i])
    end = next((i for i in range(start + 1, len(lines)) if lines[i].startswith("def ")), len(lines))
    assert a <= b < end, (func, a, b, end)
    return textwrap.dedent("\n".join(lines[a:b + 1])), (a + 1, b + 1)


def run_block(src, namespace, label):
    exec(compile(src, "<reference:%s>" % label, "exec"), namespace)
    return namespace
