"""String enums of the reference config surface (defs_.py:4-122) and the reflective resolver
`defs.check("defs.phase.train", defs.phase)` the YAML values go through (defs_.py:6-34)."""
import inspect

from .utils_ import error


class defs:
    @staticmethod
    def check(arg, should_belong_to, do_boolean=False):
        parts = str(arg).split(".")
        if parts[0] != "defs":
            if do_boolean:
                return (False, None)
            error("Invalid def : %s" % arg)
        curr, belongs_ok = defs, False
        for part in parts[1:]:
            if not belongs_ok:
                belongs_ok = should_belong_to == curr
            fields = [v[0] for v in inspect.getmembers(curr, lambda a: not inspect.isroutine(a))
                      if not (v[0].startswith("__") or v[0].endswith("__"))]
            if part not in fields:
                if do_boolean:
                    return (False, None)
                error("Parameter [%s] is not defined for [%s]" % (part, curr))
            curr = getattr(curr, part)
        if not belongs_ok:
            if do_boolean:
                return (False, None)
            error("Supplied parameter [%s] should be a child of def [%s]" % (arg, should_belong_to))
        return (True, curr) if do_boolean else curr

    class representation:
        dcnn, fc, nop = "dcnn", "fc", "nop"

    class classifier:
        fc, lstm = "fc", "lstm"

    class phase:
        train, val = "train", "val"

    class input_mode:
        video, image, vectors = "video", "image", "vectors"

    class net_input:
        visual, labels = "visual", "labels"

    class dataset_tag:
        main, aux = "main", "aux"

    class data_format:
        raw, tfrecord = "raw", "tfrecord"

    class fusion_method:
        avg, last, concat, reshape, state, ibias, maximum = "avg", "last", "concat", "reshape", "state", "ibias", "maximum"

    class fusion_type:
        early, late, none, main, aux = "early", "late", "none", "main", "aux"

    class clipframe_mode:
        rand_frames, rand_clips, iterative = "rand_frames", "rand_clips", "iterative"

    class generation_error:
        abort, compromise, report = "abort", "compromise", "report"

    class batch_item:
        default, clip = "default", "clip"

    class optim:
        sgd, rmsprop, adam = "sgd", "rmsprop", "adam"

    class decay:
        exp, staircase = "exp", "staircase"

    class periodicity:
        interval, drops = "interval", "drops"

    class names:
        global_step, latest_savefile = "global_step", "latest"

    class imgproc:
        rand_mirror, rand_crop, center_crop, resize, raw_resize, sub_mean = \
            "rand_mirror", "rand_crop", "center_crop", "resize", "raw_resize", "sub_mean"

        @staticmethod
        def to_str(vec):
            m = (("rand_mirror", "rm"), ("rand_crop", "rc"), ("center_crop", "cc"), ("resize", "rs"), ("raw_resize", "rr"),
                 ("sub_mean", "sm"))
            return "-".join(s for k, s in m if k in vec)
