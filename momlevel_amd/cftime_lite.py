"""The few calendar facts ``annual_average`` and the test-data time axes need.

The reference uses cftime (absent from this image).  ``DatetimeLite`` exposes
the attributes momlevel reads from a ``cftime.datetime`` -- ``year``, ``month``,
``day``, ``hour``, ``calendar``, ``daysinmonth`` -- so either kind of object can
sit in a time coordinate handed to ``util.annual_average``.
"""

from dataclasses import dataclass

_DPM = (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31)


def is_leap(year, calendar):
    calendar = calendar.lower()
    if calendar in ("noleap", "365_day", "360_day"):
        return False
    if calendar in ("all_leap", "366_day"):
        return True
    if calendar == "julian":
        return year % 4 == 0
    if calendar in ("standard", "gregorian", "proleptic_gregorian"):
        return (year % 4 == 0 and year % 100 != 0) or year % 400 == 0
    raise ValueError(f"unsupported calendar {calendar!r}")


def days_in_month(year, month, calendar):
    if calendar.lower() == "360_day":
        return 30
    return _DPM[month - 1] + (1 if (month == 2 and is_leap(year, calendar)) else 0)


def days_in_year(year, calendar):
    return sum(days_in_month(year, m, calendar) for m in range(1, 13))


@dataclass(frozen=True, order=True)
class DatetimeLite:
    year: int
    month: int
    day: int
    hour: int = 0
    minute: int = 0
    calendar: str = "noleap"

    @property
    def daysinmonth(self):
        return days_in_month(self.year, self.month, self.calendar)

    def __str__(self):
        return (f"{self.year:04d}-{self.month:02d}-{self.day:02d} "
                f"{self.hour:02d}:{self.minute:02d}:00")


def _from_day_of_year(year, doy, calendar):
    """doy: fractional days since Jan 1 00:00 of ``year`` (0-based)."""
    whole = int(doy)
    minutes = int(round((doy - whole) * 1440.0))
    month = 1
    while whole >= days_in_month(year, month, calendar):
        whole -= days_in_month(year, month, calendar)
        month += 1
    return DatetimeLite(year, month, whole + 1, minutes // 60, minutes % 60, calendar)


def year_midpoint(year, calendar):
    """bounds[0] + (bounds[1] - bounds[0]) / 2 of util.py:93-98 (a cftime.datetime when cftime
    is installed, so that xarray sees a CFTimeIndex as it does with the reference)."""
    mid = _from_day_of_year(year, days_in_year(year, calendar) / 2.0, calendar)
    try:
        import cftime

        return cftime.datetime(mid.year, mid.month, mid.day, mid.hour, mid.minute,
                               calendar=calendar)
    except Exception:
        return mid


def monthly_midpoints(start_year, nyears, calendar):
    """Monthly ('MS') mid-point axis of test_data/time.py:66-101."""
    out = []
    for y in range(start_year, start_year + nyears):
        doy = 0.0
        for m in range(1, 13):
            n = days_in_month(y, m, calendar)
            out.append(_from_day_of_year(y, doy + n / 2.0, calendar))
            doy += n
    return out
