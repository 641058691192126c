#!/usr/bin/env python3
"""`python wisecondor.py <newref|newrefprep|newrefpart|newrefpost|test> ...` --
the reference's entry point name, backed by the MI355X build."""
from wisecondor_amd.wisecondor import main

if __name__ == '__main__':
    main()
