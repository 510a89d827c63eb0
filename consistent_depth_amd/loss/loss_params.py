"""Loss command-line flags (same names and defaults as /root/reference/loss/loss_params.py:5-40,
including the int-typed defaults that make the run tag read `B0.1_R1.0_PL1-0`)."""

_FLAGS = (
    # flag, default, help
    ("--lambda_view_baseline", -1,
     "weight of the disparity term; < 0 selects the default of the chosen depth model"),
    ("--lambda_reprojection", 1.0, "weight of the reprojection term"),
    ("--lambda_parameter", 0, "weight of the L1 pull towards the initial network weights"),
)


class LossParams:
    @staticmethod
    def add_arguments(parser):
        for flag, default, text in _FLAGS:
            parser.add_argument(flag, type=float, default=default, help=text)
        return parser

    @staticmethod
    def make_str(opt):
        return f"B{opt.lambda_view_baseline}_R{opt.lambda_reprojection}_PL1-{opt.lambda_parameter}"
