import sys; sys.path.insert(0,'.')
from sklearn.utils.estimator_checks import check_estimator
from neo_ls_svm_amd import NeoLSSVM
for kind in ("regressor","classifier"):
    res=check_estimator(NeoLSSVM(estimator_type=kind), on_fail=None)
    for r in res:
        if r["status"]=="failed":
            print(kind, r["check_name"], "::", str(r["exception"])[:400].replace("\n"," | "))
    print(kind, "passed", sum(r["status"]=="passed" for r in res), "skipped", sum(r["status"] not in ("passed","failed") for r in res))
