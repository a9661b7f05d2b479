"""Alias of oracle/index_match.py (the kept-set checker lives with the oracle so that bench.py does not depend on tests/)."""
from oracle.index_match import count_flip_report, nlvr_index_match, set_match  # noqa: F401
