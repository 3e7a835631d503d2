#!/bin/bash
# Round-end measurement of HEAD, run from the BUILD container:   bash tools_dev/final_run.sh [tag]
#   1. refuses to run on a dirty tree (what is measured must be a commit);
#   2. writes `git rev-parse HEAD` to profiles/.measured_head (git-ignored; it travels to the GPU box with the snapshot, which
#      carries no .git) and sends tools_dev/final_box.sh to an MI355X: the full -m gpu suite, smoke(), the default bench line, the
#      other workloads, a rocprofv3 kernel trace of the bench and the PMC passes (one counter group per pass);
#   3. tools_dev/collect_profiles.py copies the summaries into profiles/round6_* -- every JSON carries "head", every CSV a first
#      line "# head <hash>", and <name>.head sits beside each file -- ready to be committed as the round's LAST commit (that commit
#      touches profiles/ and documents only: the measured tree is its parent).
set -e
tag=${1:-r6final}
root=$(cd $(dirname $0)/.. && pwd)
cd $root
if [ -n "$(git status --porcelain)" ]; then
  echo "final_run.sh: the tree is dirty -- commit first, the profiles must name the commit they measured" >&2
  git status --short >&2
  exit 1
fi
head=$(git rev-parse HEAD)
echo $head > profiles/.measured_head
python3 -c "import __graft_entry__ as g; g.build()"
/usr/local/graft/bin/gpurun --timeout 3300 -- "bash tools_dev/final_box.sh $tag"
python3 tools_dev/collect_profiles.py $tag $head
rm -f profiles/.measured_head
echo "measured $head; now: git add profiles && git commit"
