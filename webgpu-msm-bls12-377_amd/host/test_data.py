"""The harness's test-case files and known answers (row f1 of SURVEY.md section 8).

Reference: src/test-data/testCases.ts:9-52 (`getExpectedResult`, `loadTestCase`) and
src/test-data/saveTestCaseToFile.ts:1-24.  File formats:
  test-data/points/<power>-power-points.txt    one JSON object per line: {"x": "<dec>", "y": "<dec>", "z": "1"}
  test-data/scalars/<power>-power-scalars.txt  one decimal integer per line (the saver also writes
                                               the `"<dec>",` form; both are accepted)
The input files live in another repository (README.md:28-35) and are not in the tree; once they are
supplied the published 2^16..2^20 answers below become checkable end to end.
"""
import json
import os
from typing import Dict, List, Optional, Tuple

# src/test-data/testCases.ts:14-26 (data, not code)
EXPECTED_RESULTS: Dict[int, Tuple[int, int]] = {
    16: (94006842082116618334698674554269938560504658220442275405704974851793018623976750030932275315377339755327327987799, 20373698276638985490622302772174938574967913528479846848006540077491753947648956036093654307050792702539840457541),
    17: (206224560584082546776307678440614275320062113355561962308721799926405988566792861311857124914191508657092244026797, 211505771810605149801236229583532591257930087722075039263647957125630724803810862016000585191202320499088754389346),
    18: (213590253091531711003295174396041900486736230199904022674226470027355022490783453188751023812621283421365133044335, 166168294849747437548140695864136486986897221068029518430368940173172785864820517559403857089626657281214248033436),
    19: (227918075012010659569854027573177112762469117095506192259456355647196733855535622181356473956903755312919537388289, 232048820726736272000228087347068589163288439026577981179126188061989792518064409423298246183820422050991578154066),
    20: (105645455159295492078411402285457085811978509815703136952786959329738979428758249440990135440135199333488003965024, 217434031274260429359512002379640961971443333898312105830518865556255108267359047513395163712830071551228264849716),
}


def get_expected_result(power: int) -> Dict[str, int]:
    """getExpectedResult (testCases.ts:11-31); unknown powers give {x: 0, y: 0} like the reference."""
    x, y = EXPECTED_RESULTS.get(power, (0, 0))
    return {"x": x, "y": y}


def parse_points_text(text: str) -> List[Dict[str, int]]:
    pts = []
    for line in text.strip().split("\n"):
        line = line.strip()
        if not line:
            continue
        obj = json.loads(line)
        pts.append({k: int(v) for k, v in obj.items()})
    return pts


def parse_scalars_text(text: str) -> List[int]:
    out = []
    for line in text.strip().split("\n"):
        line = line.strip().rstrip(",").strip('"')
        if line:
            out.append(int(line))
    return out


def load_test_case(power: int, base_dir: str = "test-data") -> Dict[str, object]:
    """loadTestCase (testCases.ts:34-52) from a local directory instead of fetch()."""
    with open(os.path.join(base_dir, "points", "%d-power-points.txt" % power)) as f:
        points = parse_points_text(f.read())
    with open(os.path.join(base_dir, "scalars", "%d-power-scalars.txt" % power)) as f:
        scalars = parse_scalars_text(f.read())
    return {"baseAffinePoints": points, "scalars": scalars, "expectedResult": get_expected_result(power)}


def save_points_to_file(points, path: str) -> None:
    """savePointsToFile (saveTestCaseToFile.ts:1-11): one {"x","y","z"} object per line."""
    with open(path, "w") as f:
        f.write("\n".join('{ "x": "%d", "y": "%d", "z": "%d"}' % (p["x"], p["y"], p.get("z", 1)) for p in points))


def save_scalars_to_file(scalars, path: str, loader_format: bool = True) -> None:
    """saveScalarsToFile (saveTestCaseToFile.ts:13-24) writes `"<dec>",` lines; loadTestCase reads
    bare decimals.  loader_format=True writes the form the loader accepts."""
    with open(path, "w") as f:
        if loader_format:
            f.write("\n".join(str(int(s)) for s in scalars))
        else:
            f.write("\n".join('"%d",' % int(s) for s in scalars))
