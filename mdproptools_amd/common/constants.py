"""
LAMMPS unit styles -> SI factors, with the names the reference exposes
(/root/reference/mdproptools/common/constants.py:22-168): `BOLTZMANN`,
`SUPPORTED_UNITS` and one `<QUANTITY>_CONVERSION[units]` table per quantity.

The factors are physical constants (2019 SI / CODATA 2018). They are built from
the same floating-point expressions as the reference uses (e.g. 1.602176634 *
10 ** -19, which is NOT the double 1.602176634e-19), because the hot path
multiplies by them before reducing and parity is checked to ~1e-10.
"""

BOLTZMANN = 1.380649 * 10 ** -23  # J/K
ELEMENTARY_CHARGE = 1.602176634 * 10 ** -19  # C
AVOGADRO = 6.02214076 * 10 ** 23  # 1/mol
LIGHT_SPEED = 299792458  # m/s
BOHR_RADIUS = 5.29177210903 * 10 ** -11  # m
CAL_TO_J = 4.184
HA_TO_J = 4.3597447222071 * 10 ** -18

SUPPORTED_UNITS = ["real", "metal", "si", "cgs", "electron", "micro", "nano"]

_PER_MOLE_GRAM = 10 ** -3 / AVOGADRO

# base quantities per unit style: mass [kg], distance [m], time [s], energy [J], charge [C]
_BASE = {
    "real": dict(mass=_PER_MOLE_GRAM, dist=10 ** -10, time=10 ** -15,
                 energy=10 ** 3 * CAL_TO_J / AVOGADRO, charge=ELEMENTARY_CHARGE),
    "metal": dict(mass=_PER_MOLE_GRAM, dist=10 ** -10, time=10 ** -12,
                  energy=ELEMENTARY_CHARGE, charge=ELEMENTARY_CHARGE),
    "si": dict(mass=1, dist=1, time=1, energy=1, charge=1),
    "cgs": dict(mass=10 ** -3, dist=10 ** -2, time=1, energy=10 ** -7, charge=1 / 10 / LIGHT_SPEED),
    "electron": dict(mass=_PER_MOLE_GRAM, dist=BOHR_RADIUS, time=10 ** -15, energy=HA_TO_J,
                     charge=ELEMENTARY_CHARGE),
    "micro": dict(mass=10 ** -3 * 10 ** -12, dist=10 ** -6, time=10 ** -6,
                  energy=10 ** -3 * 10 ** -12, charge=10 ** -12),
    "nano": dict(mass=10 ** -3 * 10 ** -18, dist=10 ** -9, time=10 ** -9,
                 energy=10 ** -3 * 10 ** -18, charge=ELEMENTARY_CHARGE),
}


def _table(key):
    return {u: _BASE[u][key] for u in SUPPORTED_UNITS}


MASS_CONVERSION = _table("mass")
DISTANCE_CONVERSION = _table("dist")
TIME_CONVERSION = _table("time")
ENERGY_CONVERSION = _table("energy")
CHARGE_CONVERSION = _table("charge")

VELOCITY_CONVERSION = {u: DISTANCE_CONVERSION[u] / TIME_CONVERSION[u] for u in SUPPORTED_UNITS}
VELOCITY_CONVERSION["si"] = 1
VELOCITY_CONVERSION["electron"] = DISTANCE_CONVERSION["electron"] / (1.03275 * 10 ** -15)  # atomic time unit

FORCE_CONVERSION = {u: ENERGY_CONVERSION[u] / DISTANCE_CONVERSION[u] for u in SUPPORTED_UNITS}
FORCE_CONVERSION["si"] = 1
TORQUE_CONVERSION = ENERGY_CONVERSION
TEMPERATURE_CONVERSION = {u: 1 for u in SUPPORTED_UNITS}

PRESSURE_CONVERSION = {
    "real": 101325,  # atm
    "metal": 10 ** 5,  # bar
    "si": 1,
    "cgs": 10 ** -6 * 10 ** 5,  # dyne/cm^2
    "electron": 1,
    "micro": ENERGY_CONVERSION["micro"] / DISTANCE_CONVERSION["micro"] ** 3,
    "nano": ENERGY_CONVERSION["nano"] / DISTANCE_CONVERSION["nano"] ** 3,
}

VISCOSITY_CONVERSION = {
    "real": 0.1, "metal": 0.1, "si": 1, "cgs": 0.1, "electron": 1,  # poise -> Pa s
    "micro": PRESSURE_CONVERSION["micro"] * TIME_CONVERSION["micro"],
    "nano": PRESSURE_CONVERSION["nano"] * TIME_CONVERSION["nano"],
}

DIPOLE_CONVERSION = {u: CHARGE_CONVERSION[u] * DISTANCE_CONVERSION[u] for u in SUPPORTED_UNITS}
DIPOLE_CONVERSION["si"] = 1
DIPOLE_CONVERSION["electron"] = 10 ** -21 / LIGHT_SPEED  # debye

ELECTRIC_FIELD_CONVERSION = {u: 1 / DISTANCE_CONVERSION[u] for u in SUPPORTED_UNITS}
ELECTRIC_FIELD_CONVERSION["si"] = 1
ELECTRIC_FIELD_CONVERSION["cgs"] = FORCE_CONVERSION["cgs"] / CHARGE_CONVERSION["cgs"]
ELECTRIC_FIELD_CONVERSION["electron"] = 1 / 10 ** -2

_GCC = MASS_CONVERSION["cgs"] / DISTANCE_CONVERSION["cgs"] ** 3
DENSITY_3D_CONVERSION = {
    "real": _GCC, "metal": _GCC, "si": 1, "cgs": _GCC,
    "micro": MASS_CONVERSION["micro"] / DISTANCE_CONVERSION["micro"] ** 3,
    "nano": MASS_CONVERSION["nano"] / DISTANCE_CONVERSION["nano"] ** 3,
}
