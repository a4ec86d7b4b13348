"""Command line of main.py: the reference's five flags (Parser.py:4-17), same names, types
and defaults — including `type=bool`, for which any non-empty string parses as True."""
import argparse

FLAGS = (
    ("--seed_flag", bool, True, "Fix random seed or not"),
    ("--seed", int, 2024, "random seed for init"),
    ("--cuda", bool, True, "use gpu or not"),
    ("--gpu_id", int, 0, "gpu id"),
    ("--model", str, "unknown", "model name"),
)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="ID-GRec")
    for flag, kind, default, text in FLAGS:
        parser.add_argument(flag, type=kind, default=default, help=text)
    return parser.parse_args(argv)
