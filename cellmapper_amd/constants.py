"""Package constants (reference: constants.py:1-4)."""


class PackageConstants:
    n_comps: int = 50  # default number of representation columns used for the neighbor search
