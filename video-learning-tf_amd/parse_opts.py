"""parse_opts.py:6-17: tuples in the YAML are strings parsed with ast.literal_eval."""
from ast import literal_eval

from .utils_ import error


def parse_seq(arg):
    if isinstance(arg, (list, tuple)):
        return arg
    try:
        return literal_eval(arg)
    except Exception:
        error("Unable to literal-eval expression [%s]" % arg)


def to_list(arg):
    return arg if isinstance(arg, (list, tuple)) else [arg]
