"""Session API and command line of the sampler.

Mirrors numbskull/numbskull.py of the reference: the ``arguments`` / ``flags`` option tables
(18-149; same option strings, ``dest`` names and defaults -- other code iterates these tables),
class ``NumbSkull`` (152-391) with ``loadFactorGraphRaw`` / ``loadFactorGraph`` /
``loadFGFromFile`` / ``getFactorGraph`` / ``inference`` / ``learning``, and ``load`` / ``main``
(394-423).  Sampling and learning themselves run on the MI355X through
:class:`numbskull_amd.factorgraph.FactorGraph`.

Engine options that the reference does not have (``device``, ``seed``, ``scan``,
``head_by_vid``) live in ``engine_arguments`` so that the two reference tables keep their exact
shape.
"""

import argparse
import os
import sys

import numpy as np

from . import dataloading
from .factorgraph import FactorGraph
from .numbskulltypes import Meta, Weight, Variable, Factor, FactorToVar, VarToFactor


def _opt(names, dest, default, typ, metavar, text):
    return (tuple(names), {'metavar': metavar, 'dest': dest, 'default': default, 'type': typ,
                           'help': text})


def _flag(names, dest, default, text):
    return (tuple(names), {'default': default, 'dest': dest, 'action': 'store_true', 'help': text})


# (option strings, argparse keyword arguments) -- also the source of NumbSkull's keyword defaults
arguments = [
    (('directory',), {'metavar': 'DIRECTORY', 'nargs': '?', 'default': '.', 'type': str,
                      'help': 'directory holding the factor graph files'}),
    _opt(('-o', '--output_dir'), 'output_dir', '.', str, 'OUTPUT_DIR',
         'where inference_result.out.text and inference_result.out.weights.text are written'),
    _opt(('-m', '--meta', '--fg_meta'), 'metafile', 'graph.meta', str, 'META_FILE',
         'graph metadata file'),
    _opt(('-w', '--weight', '--weights'), 'weightfile', 'graph.weights', str, 'WEIGHTS_FILE',
         'weights file'),
    _opt(('-v', '--variable', '--variables'), 'variablefile', 'graph.variables', str,
         'VARIABLES_FILE', 'variables file'),
    _opt(('-f', '--factor', '--factors'), 'factorfile', 'graph.factors', str, 'FACTORS_FILE',
         'factors file'),
    _opt(('--domain', '--domains'), 'domainfile', 'graph.domains', str, 'DOMAINS_FILE',
         'categorical domains file'),
    _opt(('-l', '--n_learning_epoch'), 'n_learning_epoch', 0, int, 'NUM_LEARNING_EPOCHS',
         'number of learning epochs'),
    _opt(('-i', '--n_inference_epoch'), 'n_inference_epoch', 0, int, 'NUM_INFERENCE_EPOCHS',
         'number of inference epochs'),
    _opt(('-s', '--stepsize', '-a', '--alpha'), 'stepsize', 0.01, float, 'LEARNING_STEPSIZE',
         'learning step size'),
    _opt(('-d', '--decay', '--diminish'), 'decay', 0.95, float, 'LEARNING_DECAY',
         'step size multiplier applied after every learning epoch'),
    _opt(('-r', '--reg_param'), 'reg_param', 0.01, float, 'LEARNING_REGULARIZATION_PARAM',
         'regularization penalty'),
    _opt(('--regularization',), 'regularization', 2, int, 'REGULARIZATION',
         'regularization kind: 1 (l1, truncated gradient) or 2 (l2)'),
    _opt(('-k', '--truncation'), 'truncation', 1, int, 'TRUNCATION',
         'with l1: truncate with probability 1/k by step_size * reg_param * k'),
    _opt(('-b', '--burn_in'), 'burn_in', 0, int, 'BURN_IN', 'number of burn-in epochs'),
    _opt(('-t', '--threads', '--n_threads'), 'nthreads', 1, int, 'NUM_THREADS',
         'CPU threads of the reference sampler (accepted, ignored by the GPU engine)'),
    _opt(('-u', '--dburl'), 'dburl', '', str, 'DATABASE_URL',
         'database holding the factor graph (unused)'),
]

flags = [
    _flag(('--sample_evidence',), 'sample_evidence', True, 'sample evidence variables'),
    _flag(('--learn_non_evidence',), 'learn_non_evidence', False,
          'learn from non-evidence variables'),
    _flag(('-q', '--quiet'), 'quiet', False, 'quiet'),
    _flag(('--verbose',), 'verbose', False, 'verbose'),
]

# options of the MI355X engine (not part of the reference's tables)
engine_arguments = [
    _opt(('--device',), 'device', 0, int, 'HIP_DEVICE', 'HIP device ordinal'),
    _opt(('--seed',), 'seed', 0, int, 'SEED', 'sampler seed (Philox key / MT19937 seed)'),
    _opt(('--scan',), 'scan', 'chromatic', str, 'SCAN',
         '"chromatic" (parallel colour classes) or "sequential" (reference trajectory, slow)'),
    _opt(('--learn_cap',), 'learn_cap', 0.5, float, 'LEARN_CAP',
         'chromatic learning: cap on (visits of a weight in one colour class) x stepsize; 0 = off'),
]
engine_flags = [
    _flag(('--head_by_vid',), 'head_by_vid', False,
          'IMPLY_MLN-type factors read their head through fmap[l].vid'),
    _flag(('--no_learn_lag',), 'no_learn_lag', False,
          'chromatic learning: every colour class waits for the previous class\'s weight update '
          '(default: the update overlaps the next class, which sees weights one class older)'),
]


class NumbSkull(object):
    """A sampling session holding a list of factor graphs (numbskull.py:152-391)."""

    def __init__(self, **kwargs):
        defaults = {}
        for names, opts in arguments + engine_arguments:
            defaults['directory' if 'directory' in names[0] else opts['dest']] = opts['default']
        for names, opts in flags + engine_flags:
            defaults[opts['dest']] = opts['default']
        for name, default in defaults.items():
            setattr(self, name, kwargs.get(name, default))
        self.factorGraphs = []

    # ------------------------------------------------------------------ graph construction
    def _new_graph(self, weight, variable, factor, fmap, vmap, factor_index, var_copies,
                   weight_copies, **extra):
        fg = FactorGraph(weight, variable, factor, fmap, vmap, factor_index, var_copies,
                         weight_copies, len(self.factorGraphs), self.nthreads,
                         device=self.device, seed=self.seed, scan=self.scan, learn_cap=self.learn_cap,
                         learn_lag=not self.no_learn_lag,
                         head_by_vid=self.head_by_vid, **extra)
        self.factorGraphs.append(fg)
        return fg

    def loadFactorGraphRaw(self, weight, variable, factor, fmap, vmap, factor_index,
                           var_copies=1, weight_copies=1):
        """Graph with a caller-built inverted index (numbskull.py:183-190)."""
        self._new_graph(weight, variable, factor, fmap, vmap, factor_index, var_copies,
                        weight_copies)

    def loadFactorGraph(self, weight, variable, factor, fmap, domain_mask, edges, var_copies=1,
                        weight_copies=1, factors_to_skip=np.empty(0, np.int64), own_range=None,
                        global_ids=None):
        """In-memory graph (numbskull.py:192-243).  ``factors_to_skip`` must be sorted."""
        for arr, dt in ((weight, Weight), (variable, Variable), (factor, Factor),
                        (fmap, FactorToVar)):
            assert type(arr) == np.ndarray and arr.dtype == dt
        assert type(domain_mask) == np.ndarray and domain_mask.dtype == np.bool_
        assert type(edges) == int or type(edges) == np.int64
        assert type(factors_to_skip) == np.ndarray and factors_to_skip.dtype == np.int64

        # like the reference, the edge count is recomputed from the arities (numbskull.py:217)
        nedges = int(factor["arity"].sum() - factor["arity"][factors_to_skip].sum())
        vmap, factor_index = dataloading.new_index(variable, nedges)
        dataloading.compute_var_map(variable, factor, fmap, vmap, factor_index, domain_mask,
                                    factors_to_skip)
        extra = {} if own_range is None else {"own_range": own_range}
        if global_ids is not None:
            extra["global_ids"] = global_ids
        self._new_graph(weight, variable, factor, fmap, vmap, factor_index, var_copies,
                        weight_copies, **extra)

    def loadFGFromFile(self, directory=None, metafile=None, weightfile=None, variablefile=None,
                       factorfile=None, domainfile=None, var_copies=1, weight_copies=1):
        """Graph from DeepDive binary files (numbskull.py:245-353)."""
        if not self.directory:
            print("No factor graph specified")
            return
        directory = self.directory
        metafile = metafile or self.metafile
        weightfile = weightfile or self.weightfile
        variablefile = variablefile or self.variablefile
        factorfile = factorfile or self.factorfile
        domainfile = domainfile or self.domainfile
        show = not self.quiet
        show_records = show and self.verbose

        def path(name):
            return os.path.join(directory, name)

        # graph.meta: weights,variables,factors,edges -- DeepDive appends file paths after the
        # four counts (the reference's own test/graph.meta does), which are ignored here
        with open(path(metafile)) as f:
            fields = f.read().strip().split(",")
        meta = np.zeros((), Meta)
        for name, text in zip(Meta.names, fields[:4]):
            meta[name] = int(text)
        if show:
            print("Meta:")
            for name in Meta.names:
                print("    %-10s" % (name + ":"), meta[name])
            print()

        weight = np.zeros(int(meta["weights"]), Weight)
        dataloading.load_weights(np.fromfile(path(weightfile), np.uint8), len(weight), weight)
        if show_records:
            print("Weights:")
            for i, w in enumerate(weight):
                print("    weightId:", i)
                print("        isFixed:", w["isFixed"])
                print("        weight: ", w["initialValue"])
            print()

        variable = np.zeros(int(meta["variables"]), Variable)
        dataloading.load_variables(np.fromfile(path(variablefile), np.uint8), len(variable),
                                   variable)
        sys.stdout.flush()
        if show_records:
            print("Variables:")
            for i, v in enumerate(variable):
                print("    variableId:", i)
                print("        isEvidence:  ", v["isEvidence"])
                print("        initialValue:", v["initialValue"])
                print("        dataType:    ", v["dataType"], "(",
                      dataloading.dataType(v["dataType"]), ")")
                print("        cardinality: ", v["cardinality"])
                print()

        vmap, factor_index = dataloading.new_index(variable, int(meta["edges"]))
        print("#VTF = %s" % len(vmap))
        sys.stdout.flush()

        domain_mask = np.zeros(len(variable), np.bool_)
        if os.path.isfile(path(domainfile)) and os.stat(path(domainfile)).st_size > 0:
            dataloading.load_domains(np.fromfile(path(domainfile), np.uint8), domain_mask, vmap,
                                     variable)
            sys.stdout.flush()

        factor = np.zeros(int(meta["factors"]), Factor)
        fmap = np.zeros(int(meta["edges"]), FactorToVar)
        dataloading.load_factors(np.fromfile(path(factorfile), np.uint8), len(factor), factor,
                                 fmap, domain_mask, variable, vmap)
        sys.stdout.flush()

        dataloading.compute_var_map(variable, factor, fmap, vmap, factor_index, domain_mask)
        print("COMPLETED VMAP INDEXING")
        sys.stdout.flush()
        self._new_graph(weight, variable, factor, fmap, vmap, factor_index, var_copies,
                        weight_copies)

    def getFactorGraph(self, fgID=0):
        return self.factorGraphs[fgID]

    # ------------------------------------------------------------------ runs
    def inference(self, fgID=0, out=True):
        """Burn-in + inference epochs, then the marginals dump (numbskull.py:359-371)."""
        fg = self.factorGraphs[fgID]
        fg.inference(self.burn_in, self.n_inference_epoch, sample_evidence=self.sample_evidence,
                     diagnostics=not self.quiet)
        if out:
            fg.dump_probabilities(os.path.join(self.output_dir, "inference_result.out.text"),
                                  self.n_inference_epoch)

    def learning(self, fgID=0, out=True):
        """Burn-in + learning epochs, then the weights dump (numbskull.py:373-391)."""
        fg = self.factorGraphs[fgID]
        fg.learn(self.burn_in, self.n_learning_epoch, self.stepsize, self.decay,
                 self.regularization, self.reg_param, self.truncation,
                 diagnostics=not self.quiet, verbose=self.verbose,
                 learn_non_evidence=self.learn_non_evidence)
        if out:
            fg.dump_weights(os.path.join(self.output_dir, "inference_result.out.weights.text"))


def load(argv=None):
    """Parse the command line, build a session and load its graph (numbskull.py:394-416)."""
    if argv is None:
        argv = sys.argv[1:]
    parser = argparse.ArgumentParser(description="Runs a Gibbs sampler on an MI355X", epilog="")
    parser.add_argument("--version", action='version', version="%(prog)s 0.0",
                        help="print version number")
    for names, opts in arguments + engine_arguments + flags + engine_flags:
        parser.add_argument(*names, **opts)
    ns = NumbSkull(**vars(parser.parse_args(argv)))
    ns.loadFGFromFile()
    return ns


def main(argv=None):
    """Learning, then inference (numbskull.py:419-423)."""
    ns = load(argv)
    ns.learning()
    ns.inference()
