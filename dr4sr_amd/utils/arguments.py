"""CLI surface of the reference's run.py (utils/arguments.py:3-6): two flags, -m/--model and -d/--dataset."""
import argparse


def get_default_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="DR4SR target-model training on MI355X (dr4sr_amd)")
    p.add_argument("--model", "-m", type=str, default="SASRec", help="model class name, e.g. SASRec")
    p.add_argument("--dataset", "-d", type=str, default="amazon", help="dataset name = configs/<dataset>.yaml")
    return p
