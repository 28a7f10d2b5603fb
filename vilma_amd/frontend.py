"""`vilma <command>` entry point (reference frontend.py:14-74) for the MI355X build.

`fit` is the path this build accelerates; `sim` is the adjacent caller of the same LD operator
(SURVEY.md 8f, N4).  The reference's schema tools (make_ld_schema, check_ld_schema) are out of
scope; their output formats are what `fit` consumes, so schemas built with
the reference work unchanged.
"""
import logging
import sys
from argparse import ArgumentParser

from . import VERSION
from .sim import main as sim
from .sim import args as sim_args
from .vi_options import main as fit
from .vi_options import args as fit_args

COMMANDS = {'fit': {'cmd': fit, 'parser': fit_args},
            'sim': {'cmd': sim, 'parser': sim_args}}
_NOT_PORTED = ('make_ld_schema', 'check_ld_schema')


def build_parser():
    parser = ArgumentParser(
        description='vilma v%s (MI355X build) uses variational inference to estimate variant '
                    'effect sizes from GWAS summary data while simultaneously learning the '
                    'overall distribution of effects.' % VERSION,
        usage='vilma <command> <options>')
    subparsers = parser.add_subparsers(title='Commands', dest='command')
    for name, entry in COMMANDS.items():
        sub = entry['parser'](subparsers)
        sub.add_argument('--logfile', required=False, type=str, default='',
                         help='File to store information about the vilma run. To print to '
                              'stdout use "-". Defaults to no logging.')
        sub.add_argument('--verbose', dest='verbose', action='store_true',
                         help='Log all information (as opposed to just warnings)')
    return parser


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if argv and argv[0] in _NOT_PORTED:
        sys.exit('vilma %s is not part of the MI355X build; use the reference package for it '
                 '(its outputs are read unchanged by `vilma fit`).' % argv[0])
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.command not in COMMANDS:
        parser.print_help()
        sys.exit()
    level = 10 if args.verbose else 30
    if args.logfile == '-':
        logging.basicConfig(level=level)
    elif args.logfile:
        logging.basicConfig(filename=args.logfile, level=level)
    COMMANDS[args.command]['cmd'](args)


if __name__ == '__main__':
    main()
