"""``python -m skoots_amd --image P --pretrained-checkpoint C [--log 0-4]``: the eval flags of
the reference CLI (skoots/__main__.py:19-46, 78-98).  Training / conversion flags are out of scope."""
import argparse
import glob
import logging
import os


def main():
    parser = argparse.ArgumentParser(prog="SKOOTS (MI355X)", description="skoots parameters")
    eval_args = parser.add_argument_group("eval arguments")
    eval_args.add_argument("--image", type=str, required=True, help="path to image (or a directory of *.tif)")
    eval_args.add_argument("--pretrained-checkpoint", type=str, help="path to a pretrained skoots model")
    eval_args.add_argument("--use-cached", action="store_true",
                           help="skips model evaluation and loads previously evaluated arrays")
    eval_args.add_argument("--log", type=int, default=3,
                           help="Log Level: 0-Debug, 1-Info, 2-Warning, 3-Error, 4-Critical")
    args = parser.parse_args()
    levels = [logging.DEBUG, logging.INFO, logging.WARNING, logging.ERROR, logging.CRITICAL]
    logging.basicConfig(level=levels[args.log], format="[%(asctime)s] skoots-eval [%(levelname)s]: %(message)s")
    assert args.pretrained_checkpoint is not None, (
        "Cannot evaluate SKOOTS wihtout pretrained model. --pretrained_checkpoint must not be None")
    from skoots_amd.lib.eval import eval as sk_eval
    files = sorted(glob.glob(args.image + "/*.tif")) if os.path.isdir(args.image) else [args.image]
    for f in files:
        sk_eval(f, args.pretrained_checkpoint, used_cached_data=args.use_cached)


if __name__ == "__main__":
    main()
