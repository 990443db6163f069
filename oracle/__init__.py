"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Restatements of the reference's algorithms (and of the un-vendored third-party ops it
calls) used as the checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under drawingspinup_amd/ imports this package.
"""
