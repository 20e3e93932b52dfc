"""The driver's entry points stay importable and build() passes here (it is the round's "does it build" check)."""
import __graft_entry__ as g


def test_build_entry_point():
    g.build()
