"""experiments/*.patch are kernel variants that were built, measured and not kept (docs/EXPERIMENTS.md).  They are patches against
MOVING sources, so each names the commit whose tree it applies to (experiments/APPLIES_TO.json), and this test checks it: every patch
is listed, every listed commit exists, and `patch -p1 --dry-run` succeeds on that commit's tree (VERDICT r04 "next round" #9)."""
import glob
import json
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_experiment_patch_names_the_tree_it_applies_to():
    man = json.load(open(os.path.join(ROOT, "experiments", "APPLIES_TO.json")))["applies_to"]
    patches = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "experiments", "*.patch")) + glob.glob(os.path.join(ROOT, "experiments", "*.diff")))
    assert sorted(man) == patches, (sorted(set(patches) - set(man)), sorted(set(man) - set(patches)))
    if not os.path.isdir(os.path.join(ROOT, ".git")) or shutil.which("git") is None or shutil.which("patch") is None:
        pytest.skip("no git history here (a snapshot of the tree): the manifest is complete, the dry runs need the commits")
    bad = []
    for name, sha in man.items():
        if subprocess.run(["git", "-C", ROOT, "cat-file", "-e", sha + "^{commit}"], capture_output=True).returncode != 0:
            bad.append("%s: commit %s is not in this history" % (name, sha))
            continue
        d = tempfile.mkdtemp()
        try:
            tar = subprocess.run(["git", "-C", ROOT, "archive", sha], capture_output=True, check=True)
            subprocess.run(["tar", "-x", "-C", d], input=tar.stdout, check=True)
            r = subprocess.run(["patch", "-p1", "--dry-run", "-s", "-f", "-d", d, "-i", os.path.join(ROOT, "experiments", name)],
                               capture_output=True, text=True)
            if r.returncode != 0:
                bad.append("%s does not apply to %s: %s" % (name, sha, (r.stdout + r.stderr)[-300:]))
        finally:
            shutil.rmtree(d)
    assert not bad, "\n".join(bad)
