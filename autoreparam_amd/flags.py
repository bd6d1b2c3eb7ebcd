"""Command-line flags with the reference's names and defaults (main.py:37-113 and
program_transformations.py:44-49).  absl is not a dependency: a small parser
accepts the same ``--flag=value`` / ``--flag value`` / ``--[no]flag`` spellings,
and the values live on a module-level ``FLAGS`` object as in the reference (the
library functions take it as an explicit ``flags=`` argument with FLAGS as default).
"""
import copy

_DEFS = [
    # name, type, default, help
    ("model", str, "8schools", "Model to be used."),
    ("dataset", str, "", "Dataset to be used."),
    ("inference", str, "VI", "Inference method to be used: VI, HMCtuning, or HMC."),
    ("method", str, "CP", "Method to be used: CP, NCP, i (only if inference = HMC), cVIP, dVIP."),
    ("learnable_parameterisation_type", str, "eig", "Type of learnable parameterisation (scalar models ignore it)."),
    ("reparameterise_variational", bool, False, "Not supported by this build (off in the reference too)."),
    ("discrete_prior", bool, False, "Prior encouraging the parameterisation parameters to be 0 or 1."),
    ("tied_pparams", bool, True, "Tie the loc and scale parameterisation parameters (a=b)."),
    ("results_dir", str, "", "Directory to write results."),
    ("learning_rates", list, [0.02, 0.05, 0.1, 0.2, 0.4], "Learning rates (list)"),
    ("num_optimization_steps", int, 3000, "Number of steps to optimize the ELBO."),
    ("num_mc_samples", int, 256, "Number of Monte Carlo samples to use in the ELBO."),
    ("num_leapfrog_steps", int, None, "Number of leapfrog steps."),
    ("count_in_leapfrog_steps", bool, False, "Interpret sample/burn-in/adaptation counts as gradient evaluations."),
    ("num_samples", int, 50000, "Number of HMC samples."),
    ("num_chains", int, 100, "Number of HMC chains."),
    ("num_burnin_steps", int, 10000, "Number of warm-up steps."),
    ("num_adaptation_steps", int, 6000, "Number of adaptation steps."),
    ("num_chains_to_save", int, 0, "Number of chains to save traces for."),
    ("float64", bool, False, "Unused by the scalar-Normal models of this build."),
    # build-specific additions (the reference is unseeded and single-device)
    ("seed", int, 0, "Seed of the sampler's counter-based RNG and of the variational initial states."),
    ("device", str, "cuda:0", "GPU to run on."),
    ("trace_chunk_rows", int, None, "Force the streaming mode (moments and batch means accumulated inside the kernels, "
                                    "a whole trace only for --ess_chains chains; batch length of the batch-means ESS = "
                                    "this / 8).  Default: automatic, only "
                                    "when the trace does not fit in HBM."),
    ("ess_chains", int, 1024, "Streaming mode: the chains with global id below this keep their whole [S, k, D] trace on "
                              "the device, and the reported ESS is the reference's autocorrelation ESS of those chains "
                              "(the batch-means figure of all chains is written next to it)."),
    ("lanes_per_chain", int, 0, "Lanes of a wave64 a chain is spread over (0 = automatic)."),
]


class FlagValues(object):
    def __init__(self):
        for name, _, default, _ in _DEFS:
            setattr(self, name, copy.copy(default))

    def copy(self):
        return copy.deepcopy(self)

    def parse(self, argv):
        """Parse ``argv`` (without the program name); returns the non-flag leftovers."""
        types = {n: t for n, t, _, _ in _DEFS}
        rest = []
        i = 0
        while i < len(argv):
            arg = argv[i]
            i += 1
            if not arg.startswith("--"):
                rest.append(arg)
                continue
            body = arg[2:]
            if "=" in body:
                name, val = body.split("=", 1)
            else:
                name, val = body, None
            if name not in types and name.startswith("no") and types.get(name[2:]) is bool and val is None:
                setattr(self, name[2:], False)
                continue
            if name not in types:
                raise ValueError("Unknown command line flag '%s'" % name)
            t = types[name]
            if t is bool:
                if val is None:
                    setattr(self, name, True)
                else:
                    if val.lower() in ("true", "t", "1", "yes"):
                        setattr(self, name, True)
                    elif val.lower() in ("false", "f", "0", "no"):
                        setattr(self, name, False)
                    else:
                        raise ValueError("flag --%s: bad boolean %r" % (name, val))
                continue
            if val is None:
                if i >= len(argv):
                    raise ValueError("flag --%s needs a value" % name)
                val = argv[i]
                i += 1
            if t is list:
                setattr(self, name, [v for v in val.split(",") if v != ""])
            elif t is int:
                setattr(self, name, int(val))
            else:
                setattr(self, name, val)
        return rest


FLAGS = FlagValues()
